// Do VALU and MFMA instructions of DIFFERENT waves on the same SIMD overlap on gfx950?
// One workgroup of 8 waves per CU (2 per SIMD).  Modes: 0 = every wave runs the MFMA loop,
// 1 = every wave runs the VALU loop, 2 = waves 0-3 MFMA + waves 4-7 VALU (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, unsigned long long* cyc) {
  const int wv = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && wv < 4);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float res = 0.f;
  if (mode >= 3) {
  } else if (do_mfma) {
    f32x16 a0 = {}, a1 = {};
    f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
      }
    }
    res = a0[0] + a1[3];
  } else {
    f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {  // 128 independent-ish packed fmas per iteration
        v0 = __builtin_elementwise_fma(v0, s, c);
        v1 = __builtin_elementwise_fma(v1, s, c);
        v2 = __builtin_elementwise_fma(v2, s, c);
        v3 = __builtin_elementwise_fma(v3, s, c);
      }
    }
    res = (float)(v0[0] + v1[1] + v2[0] + v3[1]);
  }
  if (mode >= 3) {
    // every wave: 4 x (13 dependent-ish packed VALU ops -> 2 MFMAs fed by them), like the GEMM consumer.
    // mode 4: the second wave of each SIMD (waves 4-7) starts half a step late.
    f32x16 a0 = {}, a1 = {};
    f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
    f16x8 y = {1, 1, 1, 1, 1, 1, 1, 1};
    if (mode == 4 && wv >= 4) __builtin_amdgcn_s_sleep(1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);
        v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);
        v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);
        v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);
        v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);
        v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);
        v0 = __builtin_elementwise_fma(v0, s, c);
        f16x8 x;
        x[0] = v0[0]; x[1] = v0[1]; x[2] = v1[0]; x[3] = v1[1]; x[4] = v2[0]; x[5] = v2[1]; x[6] = v3[0]; x[7] = v3[1];
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
      }
    }
    res = a0[0] + a1[3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
  if (res == 12345.f) out[0] = res;
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  unsigned long long h[8];
  const int iters = 2000;
  for (int mode = 0; mode < 5; ++mode) {
    k<<<256, 512>>>(mode, iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[] = {"all waves MFMA (16 x 32x32x16 per iter)", "all waves VALU (128 v_pk_fma_f16 per iter)", "waves 0-3 MFMA + 4-7 VALU", "all waves 4x(13 VALU -> 2 MFMA), lock-step", "same, second wave per SIMD offset"};
    printf("%-46s cycles/iter: wave0 %.1f  wave4 %.1f\n", nm[mode], (double)h[0] / iters, (double)h[4] / iters);
  }
  return 0;
}
