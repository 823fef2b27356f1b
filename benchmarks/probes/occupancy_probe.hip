// Does the consumer's unit of work run faster as FOUR waves per SIMD than as two?
// One "unit" per workgroup-iteration = 64 MFMA 32x32x16 f16 + 416 packed-f16 VALU + the operand reads of the
// decode GEMM (x fragments b128, weight words, scale pairs), one s_barrier per unit -- dealt to
//   NW = 8  waves (2 per SIMD): 4 steps x (13 VALU -> 2 MFMA), 8 + 2 LDS reads per wave   (today's kernel)
//   NW = 16 waves (4 per SIMD): 2 steps x (13 VALU -> 2 MFMA), 4 + 2 LDS reads per wave
// Operand reads are issued one unit ahead, right behind each step's MFMAs (the kernel's schedule), and the
// loop is unrolled by two so that the fragment buffers swap roles without register copies.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

template <int NW>
__global__ __launch_bounds__(NW * 64) void k(int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int ST = 32 / NW;  // steps per wave per unit: 4 or 2
  constexpr int NR = 2 * ST + 2;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 a0 = {}, a1 = {};
  f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
  const unsigned char* rp = lds + (size_t)(wv * 64 + lane) * 16;
  i32x4v f[NR], g[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) f[j] = *reinterpret_cast<const i32x4v*>(rp + j * (NW * 1024));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define HALF(CUR, NXT, PAR)                                                                                   \
  _Pragma("unroll") for (int st = 0; st < ST; ++st) {                                                         \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                           \
      v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);                     \
      v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);                     \
    }                                                                                                         \
    v0 = __builtin_elementwise_fma(v0, s, c);                                                                 \
    f16x8 x;                                                                                                  \
    x[0] = v0[0]; x[1] = v0[1]; x[2] = v1[0]; x[3] = v1[1]; x[4] = v2[0]; x[5] = v2[1]; x[6] = v3[0]; x[7] = v3[1]; \
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, CUR[2 * st]), a0, 0, 0, 0);      \
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, CUR[2 * st + 1]), a1, 0, 0, 0);  \
    NXT[2 * st] = *reinterpret_cast<const i32x4v*>(rp + (2 * st) * (NW * 1024) + (PAR));                      \
    NXT[2 * st + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * st + 1) * (NW * 1024) + (PAR));              \
    if (st == 0) {                                                                                            \
      NXT[2 * ST] = *reinterpret_cast<const i32x4v*>(rp + (2 * ST) * (NW * 1024));                            \
      NXT[2 * ST + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * ST + 1) * (NW * 1024));                    \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }                                                                                                           \
  v0[0] += (_Float16)(float)((CUR[2 * ST][0] ^ CUR[2 * ST + 1][1]) & 1);                                      \
  __builtin_amdgcn_s_barrier();
  for (int i = 0; i < iters; i += 2) {
    HALF(f, g, 16)
    HALF(g, f, 0)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 16 + wv] = t1 - t0;
  if (a0[0] + a1[3] == 12345.f) out[0] = a0[0];
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  unsigned long long h[16];
  const int iters = 4000;
  (void)hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    k<8><<<256, 512, 8 * 1024 * 10 + 1024>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf(" 8 consumer waves (2/SIMD), 4 steps each: %.1f cycles per unit (wave 0), %.1f (wave 7)\n", (double)h[0] / iters, (double)h[7] / iters);
    k<16><<<256, 1024, 16 * 1024 * 6 + 1024>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("16 consumer waves (4/SIMD), 2 steps each: %.1f cycles per unit (wave 0), %.1f (wave 15)\n", (double)h[0] / iters, (double)h[15] / iters);
  }
  return 0;
}
