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
  if (mode == 3 || mode == 4) {
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
  if (mode >= 5 && mode <= 7) {
    // consumer-like loop, built up in stages:  5 = mode 3 + 10 ds_read_b128 per iteration (prefetched one
    // iteration ahead), 6 = 5 + one s_barrier per iteration (8 waves), 7 = 6 with the reads of the
    // NEXT iteration issued right after each MFMA pair (the kernel's schedule)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {}, a1 = {};
    f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
    const unsigned char* rp = lds + (size_t)(wv * 64 + lane) * 16;
    i32x4v f[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) f[j] = *reinterpret_cast<const i32x4v*>(rp + j * 8192);
    for (int i = 0; i < iters; ++i) {
      i32x4v fn[10];
      if (mode != 7) {
#pragma unroll
        for (int j = 0; j < 10; ++j) fn[j] = *reinterpret_cast<const i32x4v*>(rp + j * 8192 + ((i & 1) << 4));
      }
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
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, f[2 * st]), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, f[2 * st + 1]), a1, 0, 0, 0);
        if (mode == 7) {
          fn[2 * st] = *reinterpret_cast<const i32x4v*>(rp + (2 * st) * 8192 + ((i & 1) << 4));
          fn[2 * st + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * st + 1) * 8192 + ((i & 1) << 4));
          if (st == 0) {
            fn[8] = *reinterpret_cast<const i32x4v*>(rp + 8 * 8192);
            fn[9] = *reinterpret_cast<const i32x4v*>(rp + 9 * 8192);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      v0[0] += (_Float16)(float)(f[8][0] & 1) + (_Float16)(float)(f[9][1] & 1);
      if (mode >= 6) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 10; ++j) f[j] = fn[j];
    }
    res = a0[0] + a1[3];
  }
  if (mode == 8 && wv < 4) {
    // alternative consumer shape: ONE consumer wave per SIMD owning 64 rows x 64 k of the unit:
    // per iteration 4 x (26 VALU -> 4 MFMA on 4 accumulators) + 12 LDS b128 reads (each x fragment now
    // feeds two MFMAs) + s_barrier (mode 9: the idle second wave per SIMD also arrives at the barrier)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
    f16x2 u0 = {1, 2}, u1 = {3, 4}, u2 = {5, 6}, u3 = {7, 8};
    const unsigned char* rp = lds + (size_t)(wv * 64 + lane) * 16;
    i32x4v f[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) f[j] = *reinterpret_cast<const i32x4v*>(rp + j * 4096);
    for (int i = 0; i < iters; ++i) {
      i32x4v fn[12];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);
          v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);
          u0 = __builtin_elementwise_fma(u0, s, c); u1 = __builtin_elementwise_fma(u1, s, c);
          u2 = __builtin_elementwise_fma(u2, s, c); u3 = __builtin_elementwise_fma(u3, s, c);
        }
        v0 = __builtin_elementwise_fma(v0, s, c);
        u0 = __builtin_elementwise_fma(u0, s, c);
        f16x8 x, y;
        x[0] = v0[0]; x[1] = v0[1]; x[2] = v1[0]; x[3] = v1[1]; x[4] = v2[0]; x[5] = v2[1]; x[6] = v3[0]; x[7] = v3[1];
        y[0] = u0[0]; y[1] = u0[1]; y[2] = u1[0]; y[3] = u1[1]; y[4] = u2[0]; y[5] = u2[1]; y[6] = u3[0]; y[7] = u3[1];
        const f16x8 b0 = __builtin_bit_cast(f16x8, f[2 * st]), b1 = __builtin_bit_cast(f16x8, f[2 * st + 1]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b1, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b0, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b1, a3, 0, 0, 0);
        fn[2 * st] = *reinterpret_cast<const i32x4v*>(rp + (2 * st) * 4096 + ((i & 1) << 4));
        fn[2 * st + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * st + 1) * 4096 + ((i & 1) << 4));
        if (st == 0) {
#pragma unroll
          for (int j = 8; j < 12; ++j) fn[j] = *reinterpret_cast<const i32x4v*>(rp + j * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      v0[0] += (_Float16)(float)((f[8][0] ^ f[9][1] ^ f[10][2] ^ f[11][3]) & 1);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 12; ++j) f[j] = fn[j];
    }
    res = a0[0] + a1[3] + a2[1] + a3[2];
  } else if (mode == 8) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();  // the other four waves only keep the barrier count
  }
  if (mode == 9 || (mode == 10 && wv < 4)) {
    // copy-free versions of modes 7 / 8: the loop is unrolled by two and the fragment buffers swap roles
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x2 v0 = {1, 2}, v1 = {3, 4}, v2 = {5, 6}, v3 = {7, 8}, s = {(_Float16)0.999f, (_Float16)1.001f}, c = {(_Float16)0.001f, (_Float16)-0.001f};
    f16x2 u0 = {1, 2}, u1 = {3, 4}, u2 = {5, 6}, u3 = {7, 8};
    const unsigned char* rp = lds + (size_t)(wv * 64 + lane) * 16;
    i32x4v f[12], g[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) f[j] = *reinterpret_cast<const i32x4v*>(rp + j * 4096);
#define PROBE_HALF(CUR, NXT, PAR)                                                                          \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                                       \
      if (mode == 10) {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                      \
          v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);               \
          v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);               \
          u0 = __builtin_elementwise_fma(u0, s, c); u1 = __builtin_elementwise_fma(u1, s, c);               \
          u2 = __builtin_elementwise_fma(u2, s, c); u3 = __builtin_elementwise_fma(u3, s, c);               \
        }                                                                                                    \
        v0 = __builtin_elementwise_fma(v0, s, c); u0 = __builtin_elementwise_fma(u0, s, c);                 \
      } else {                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                      \
          v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);               \
          v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);               \
        }                                                                                                    \
        v0 = __builtin_elementwise_fma(v0, s, c);                                                            \
      }                                                                                                      \
      f16x8 x, y;                                                                                            \
      x[0] = v0[0]; x[1] = v0[1]; x[2] = v1[0]; x[3] = v1[1]; x[4] = v2[0]; x[5] = v2[1]; x[6] = v3[0]; x[7] = v3[1]; \
      y[0] = u0[0]; y[1] = u0[1]; y[2] = u1[0]; y[3] = u1[1]; y[4] = u2[0]; y[5] = u2[1]; y[6] = u3[0]; y[7] = u3[1]; \
      const f16x8 b0 = __builtin_bit_cast(f16x8, CUR[2 * st]), b1 = __builtin_bit_cast(f16x8, CUR[2 * st + 1]); \
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b0, a0, 0, 0, 0);                                       \
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b1, a1, 0, 0, 0);                                       \
      if (mode == 10) {                                                                                      \
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b0, a2, 0, 0, 0);                                     \
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b1, a3, 0, 0, 0);                                     \
      }                                                                                                      \
      NXT[2 * st] = *reinterpret_cast<const i32x4v*>(rp + (2 * st) * 4096 + (PAR));                          \
      NXT[2 * st + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * st + 1) * 4096 + (PAR));                  \
      if (st == 0) {                                                                                         \
        NXT[8] = *reinterpret_cast<const i32x4v*>(rp + 8 * 4096);                                            \
        NXT[9] = *reinterpret_cast<const i32x4v*>(rp + 9 * 4096);                                            \
        if (mode == 10) {                                                                                    \
          NXT[10] = *reinterpret_cast<const i32x4v*>(rp + 10 * 4096);                                        \
          NXT[11] = *reinterpret_cast<const i32x4v*>(rp + 11 * 4096);                                        \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    v0[0] += (_Float16)(float)((CUR[8][0] ^ CUR[9][1]) & 1);                                                 \
    if (mode == 10) v0[1] += (_Float16)(float)((CUR[10][2] ^ CUR[11][3]) & 1);                               \
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < iters; i += 2) {
      PROBE_HALF(f, g, 16)
      PROBE_HALF(g, f, 0)
    }
    res = a0[0] + a1[3] + a2[1] + a3[2];
  } else if (mode == 10) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
  if (res == 12345.f) out[0] = res;
}

int main() {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  unsigned long long h[8];
  const int iters = 2000;
  for (int mode = 0; mode < 11; ++mode) {
    k<<<256, 512, 96 * 1024>>>(mode, iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[] = {"all waves MFMA (16 x 32x32x16 per iter)", "all waves VALU (128 v_pk_fma_f16 per iter)", "waves 0-3 MFMA + 4-7 VALU", "all waves 4x(13 VALU -> 2 MFMA), lock-step", "same, second wave per SIMD offset", "mode 3 + 10 LDS b128 reads/iter (one iter ahead)", "+ s_barrier per iteration", "+ reads issued behind each MFMA pair", "ONE consumer wave / SIMD: 4x(26 VALU -> 4 MFMA), 12 reads, barrier", "two waves / SIMD, copy-free (10 reads, barrier)", "one wave / SIMD, copy-free (12 reads, barrier)"};
    printf("%-46s cycles/iter: wave0 %.1f  wave4 %.1f\n", nm[mode], (double)h[0] / iters, (double)h[4] / iters);
  }
  return 0;
}
