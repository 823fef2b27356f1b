// What ds_read_b64_tr_b16 returns: LDS[i] = i (16-bit), lane l reads at element address 4*l (its own 8 bytes).
// Prints, per lane, the four 16-bit values it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
