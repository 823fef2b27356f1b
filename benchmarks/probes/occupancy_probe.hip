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

// READS: operand reads in the loop; BAR: s_barrier per unit; PIN: sched_barrier after each step
// PRIO: 0 none; 1 the younger wave of each SIMD (wv >= NW/2) runs at priority 1; 2 the older one does;
//       3 priorities swap every unit
// MID: the barrier sits in the MIDDLE of the unit (after step ST/2 - 1) and all operand reads of the next unit are issued
//      in the first half, so that what follows the barrier is pure register arithmetic
// ROT: the younger half of the waves runs the step as [MFMA(fragment dequantised earlier) ; reads ; dequantise the next]
//      so that right after a barrier one wave of the SIMD starts on the matrix pipe and the other on the VALU
template <int NW, bool READS = true, bool BAR = true, bool PIN = true, int PRIO = 0, bool MID = false, int BEVERY = 1, int ROT = 0>
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
  if (PRIO == 1 && wv >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && wv < NW / 2) __builtin_amdgcn_s_setprio(1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const bool rot = ROT == 1 ? wv >= NW / 2 : (ROT == 2 ? (wv & 1) : false);
  f16x8 xq[ST];
#pragma unroll
  for (int st = 0; st < ST; ++st) { xq[st][0] = v0[0]; xq[st][1] = v1[1]; xq[st][2] = v2[0]; xq[st][3] = v3[1]; xq[st][4] = v0[1]; xq[st][5] = v1[0]; xq[st][6] = v2[1]; xq[st][7] = v3[0]; }
  f16x8 xpre;
  xpre[0] = v0[0]; xpre[1] = v0[1]; xpre[2] = v1[0]; xpre[3] = v1[1]; xpre[4] = v2[0]; xpre[5] = v2[1]; xpre[6] = v3[0]; xpre[7] = v3[1];
#define DEQ(X)                                                                                                \
  {                                                                                                           \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                           \
      v0 = __builtin_elementwise_fma(v0, s, c); v1 = __builtin_elementwise_fma(v1, s, c);                     \
      v2 = __builtin_elementwise_fma(v2, s, c); v3 = __builtin_elementwise_fma(v3, s, c);                     \
    }                                                                                                         \
    v0 = __builtin_elementwise_fma(v0, s, c);                                                                 \
    X[0] = v0[0]; X[1] = v0[1]; X[2] = v1[0]; X[3] = v1[1]; X[4] = v2[0]; X[5] = v2[1]; X[6] = v3[0]; X[7] = v3[1]; \
  }
#define RD(CUR, NXT, PAR, st)                                                                                 \
    if constexpr (READS) {                                                                                    \
      NXT[2 * st] = *reinterpret_cast<const i32x4v*>(rp + (2 * st) * (NW * 1024) + (PAR));                    \
      NXT[2 * st + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * st + 1) * (NW * 1024) + (PAR));            \
      if (st == 0) {                                                                                          \
        NXT[2 * ST] = *reinterpret_cast<const i32x4v*>(rp + (2 * ST) * (NW * 1024));                          \
        NXT[2 * ST + 1] = *reinterpret_cast<const i32x4v*>(rp + (2 * ST + 1) * (NW * 1024));                  \
      }                                                                                                       \
    } else {                                                                                                  \
      NXT[2 * st] = CUR[2 * st]; NXT[2 * st + 1] = CUR[2 * st + 1];                                           \
      if (st == 0) { NXT[2 * ST] = CUR[2 * ST]; NXT[2 * ST + 1] = CUR[2 * ST + 1]; }                          \
    }
#define HALF(CUR, NXT, PAR)                                                                                   \
  if (ROT == 3) {                                                                                             \
    if (wv >= NW / 2) {                                                                                       \
      _Pragma("unroll") for (int st = 0; st < ST; ++st) {                                                     \
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[st], __builtin_bit_cast(f16x8, CUR[2 * st]), a0, 0, 0, 0); \
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[st], __builtin_bit_cast(f16x8, CUR[2 * st + 1]), a1, 0, 0, 0); \
        RD(CUR, NXT, PAR, st)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }                                                                                                       \
      _Pragma("unroll") for (int st = 0; st < ST; ++st) { DEQ(xq[st]) __builtin_amdgcn_sched_barrier(0); }    \
    } else {                                                                                                  \
      _Pragma("unroll") for (int st = 0; st < ST; ++st) { DEQ(xq[st]) __builtin_amdgcn_sched_barrier(0); }    \
      _Pragma("unroll") for (int st = 0; st < ST; ++st) {                                                     \
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[st], __builtin_bit_cast(f16x8, CUR[2 * st]), a0, 0, 0, 0); \
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[st], __builtin_bit_cast(f16x8, CUR[2 * st + 1]), a1, 0, 0, 0); \
        RD(CUR, NXT, PAR, st)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }                                                                                                       \
    }                                                                                                         \
  } else if (ROT && rot) {                                                                                           \
    _Pragma("unroll") for (int st = 0; st < ST; ++st) {                                                       \
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xpre, __builtin_bit_cast(f16x8, CUR[2 * st]), a0, 0, 0, 0); \
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xpre, __builtin_bit_cast(f16x8, CUR[2 * st + 1]), a1, 0, 0, 0); \
      RD(CUR, NXT, PAR, st)                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      DEQ(xpre)                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }                                                                                                         \
  } else {                                                                                                    \
    _Pragma("unroll") for (int st = 0; st < ST; ++st) {                                                       \
      f16x8 x;                                                                                                \
      DEQ(x)                                                                                                  \
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, CUR[2 * st]), a0, 0, 0, 0);    \
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(f16x8, CUR[2 * st + 1]), a1, 0, 0, 0); \
      RD(CUR, NXT, PAR, st)                                                                                   \
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                                         \
  }                                                                                                           \
  v0[0] += (_Float16)(float)((CUR[2 * ST][0] ^ CUR[2 * ST + 1][1]) & 1);                                      \
  if constexpr (BAR) __builtin_amdgcn_s_barrier();
  if constexpr (BEVERY > 1) {
    // one barrier per BEVERY units (BEVERY even): the HALF macro's own barrier is compiled out (BAR = false)
    for (int i = 0; i < iters; i += BEVERY) {
#pragma unroll
      for (int u = 0; u < BEVERY; u += 2) {
        HALF(f, g, 16)
        HALF(g, f, 0)
      }
      __builtin_amdgcn_s_barrier();
    }
  } else
  for (int i = 0; i < iters; i += 2) {
    if (PRIO == 3) { if (wv >= NW / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    HALF(f, g, 16)
    if (PRIO == 3) { if (wv >= NW / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
    HALF(g, f, 0)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 16 + wv] = t1 - t0;
  if (a0[0] + a1[3] == 12345.f) out[0] = a0[0];
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void kbar(int iters, float* out, unsigned long long* cyc) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  unsigned long long h[16];
  const int iters = 4000;
  (void)hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void)hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  auto run = [&](const char* name, auto kern, int nw) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    kern<<<256, nw * 64, 100 * 1024>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0, mn = ~0ull;
    for (int w = 0; w < nw; ++w) { if (h[w] > mx) mx = h[w]; if (h[w] < mn) mn = h[w]; }
    printf("%-58s %.1f cycles per unit (slowest wave; fastest %.1f)\n", name, (double)mx / iters, (double)mn / iters);
  };
  run("2/SIMD: arithmetic only (no reads, no barrier)", k<8, false, false, true>, 8);
  run("2/SIMD: + barrier", k<8, false, true, true>, 8);
  run("2/SIMD: + reads, no barrier", k<8, true, false, true>, 8);
  run("2/SIMD: reads + barrier, compiler-scheduled (no pins)", k<8, true, true, false>, 8);
  run("2/SIMD: reads + barrier, pinned (the kernel's schedule)", k<8, true, true, true>, 8);
  run("2/SIMD: reads + barrier, younger wave at priority 1", k<8, true, true, true, 1>, 8);
  run("2/SIMD: reads + barrier, older wave at priority 1", k<8, true, true, true, 2>, 8);
  run("2/SIMD: reads + barrier, priorities swap every unit", k<8, true, true, true, 3>, 8);
  run("2/SIMD: barrier only, younger wave at priority 1", k<8, false, true, true, 1>, 8);
  run("2/SIMD: reads in the first half + barrier MID-unit", k<8, true, true, true, 0, true>, 8);
  run("4/SIMD: reads in the first half + barrier MID-unit", k<16, true, true, true, 0, true>, 16);
  run("2/SIMD: reads, one barrier per 2 units", k<8, true, false, true, 0, false, 2>, 8);
  run("2/SIMD: reads, one barrier per 4 units", k<8, true, false, true, 0, false, 4>, 8);
  run("4/SIMD: reads, one barrier per 2 units", k<16, true, false, true, 0, false, 2>, 16);
  run("4/SIMD: reads, one barrier per 4 units", k<16, true, false, true, 0, false, 4>, 16);
  run("2/SIMD: reads + barrier, younger wave software-rotated", k<8, true, true, true, 0, false, 1, 1>, 8);
  run("2/SIMD: reads, no barrier, younger wave rotated", k<8, true, false, true, 0, false, 1, 1>, 8);
  run("4/SIMD: reads + barrier, odd waves rotated", k<16, true, true, true, 0, false, 1, 2>, 16);
  run("4/SIMD: reads + barrier, younger half rotated", k<16, true, true, true, 0, false, 1, 1>, 16);
  run("2/SIMD: reads + barrier, BLOCK anti-phase (A: dequant-all then MFMA-all, B: reverse)", k<8, true, true, true, 0, false, 1, 3>, 8);
  run("2/SIMD: barrier, no reads, block anti-phase", k<8, false, true, true, 0, false, 1, 3>, 8);
  run("4/SIMD: reads + barrier, block anti-phase", k<16, true, true, true, 0, false, 1, 3>, 16);
  run("bare s_barrier loop,  4 waves (1/SIMD)", kbar<4>, 4);
  run("bare s_barrier loop,  8 waves (2/SIMD)", kbar<8>, 8);
  run("bare s_barrier loop, 12 waves (3/SIMD)", kbar<12>, 12);
  run("bare s_barrier loop, 16 waves (4/SIMD)", kbar<16>, 16);
  run("4/SIMD: arithmetic only", k<16, false, false, true>, 16);
  run("4/SIMD: + barrier", k<16, false, true, true>, 16);
  run("4/SIMD: + reads, no barrier", k<16, true, false, true>, 16);
  run("4/SIMD: reads + barrier, pinned", k<16, true, true, true>, 16);
  for (int rep = 0; rep < 1; ++rep) {
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
