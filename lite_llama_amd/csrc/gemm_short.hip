// W4A16 decode GEMM, SHORT-STREAM engine (round 6): split-K partial launches whose whole weight stream is a few tens of KB per CU
// (the fused q|k|v and the attention output projection of a 7B model: 9.3 / 7.2 MB = 36 / 28 KB per CU; every projection of a
// TP >= 4 shard).  Semantics: lite_llama/kernels/quantization/w4a16.py:152-207 (out[m, n] = sum_k x[m, k] (nib(n, k) - z) s, fp32
// accumulation), left as S fp32 split-K planes [S][M][N] for the consumer of the projection (ll_skip_rmsnorm_partials,
// ll_decode_attention_partials) exactly like epilogue 2 of the unit loop (gemm_w4_v3.hip), over the SAME load-time layouts.
//
// The unit loop and the row-group loop are pipelines built for 40 - 76 MB streams: rings, loader / consumer hand-offs per unit,
// 3 us to the first finished unit and ~1 us of tail.  A launch that moves ONE ring's worth of bytes per CU is all ramp and tail
// there (8.6 - 10.1 us for 7 - 9 MB).  This body is the opposite shape:
//   * one workgroup per CU, work item = (R row groups of 32 weight rows, one k-slice); no ring, no unit loop;
//   * the 8 consumer waves request ALL their weight words in their first ~100 cycles, straight into registers (1 KB per wave
//     instruction = one (row group, 64-k block) piece of the packed layout, non-temporal), and the (s, -z s) pairs behind them;
//   * 4 loader waves bring the k-slice of the activation matrix into LDS ONCE (LDS-DMA, swizzled through the source address as in
//     the other engines) -- the workgroups of one XCD all work on one or two k-slices, so an XCD's L2 sees 1 / S of the matrix;
//   * a consumer dequantises + multiplies a piece as soon as IT has landed (the compiler's counted vmcnt: loads return in order);
//     the activation slice arrives in two stages so the first pieces do not wait for its tail;
//   * wave (r, q) = row group r of the item x k-blocks q, q + KQ, ...; the KQ partial sums of a row group meet once through LDS,
//     in a fixed order (deterministic), and leave as full 128-byte lines of the plane.
#include <stdlib.h>

#include "common.h"
#include "gemm_debug.h"  // SS_TL / SS_TLC: in-kernel stamps of -DSS_TIMELINE builds (empty in the product build)
#include "gemm_w4_common.h"

#define SS_CONSUMERS 8
#ifndef SS_LOADERS
#define SS_LOADERS 4  // (-DSS_LOADERS=8: A/B build, 16-wave workgroups)
#endif
#define SS_THREADS ((SS_CONSUMERS + SS_LOADERS) * 64)
#define SS_MAX_CHUNKS 9      // activation chunks (128 k) of one k-slice resident in LDS
#define SS_MAX_SLICES 8      // planes the consumers of a projection add up (flash_decoding.hip FD_QS_MAX)

// The kernel takes PLAIN arguments, the ones the first weight request needs first: with -mllvm -amdgpu-kernarg-preload-count=16
// (lite_llama_amd/build.py, this file only) the first 16 dwords arrive in SGPRs with the wave instead of through three dependent
// scalar-load round trips (~0.3 us of a ~5-us launch).
struct SSParams {
  const char* wp;      // packed weights (ll_w4a16_pack_weights)
  const char* sp;      // packed (s, -z s) pairs [K / g][N] x 8 B
  int n;
  int chunks;          // K / 128
  int gshift;          // log2(group_size / 128)
  int kb_base, kb_rem; // k-slice s owns the 64-k blocks [s kb_base + min(s, kb_rem), + kb_base + (s < kb_rem))
  int RB;              // row blocks (R row groups each)
  uint32_t rb_magic;   // ceil(2^32 / RB) (0: RB = 1): item / RB = umulhi(item, rb_magic)
  int total;           // work items = S * RB, item = slice * RB + row block
  int cap;             // work items per XCD: workgroup b (XCD b % 8 under the round-robin dispatch) takes item (b % 8) cap + b / 8
  int m;
  const uint16_t* x;   // [M][K] fp16, row stride x_stride elements
  float* out;          // [S][M][N] fp32
  int x_stride;
  unsigned long long* tl;  // SS_TIMELINE builds: [workgroup][wave 12][16] stamps (benchmarks/gemm_short_timeline.py); else unused
};
#define SS_ARGS const char* __restrict__ p_wp, const char* __restrict__ p_sp, int p_n, int p_chunks, int p_gshift, int p_kb_base, \
                int p_kb_rem, int p_RB, uint32_t p_rb_magic, int p_total, int p_cap, int p_m, const uint16_t* __restrict__ p_x, \
                float* __restrict__ p_out, int p_x_stride, unsigned long long* p_tl
#define SS_PASS(P) P.wp, P.sp, P.n, P.chunks, P.gshift, P.kb_base, P.kb_rem, P.RB, P.rb_magic, P.total, P.cap, P.m, P.x, P.out, P.x_stride, P.tl


__device__ __forceinline__ void ss_barrier() { asm volatile("s_barrier" ::: "memory"); }

// at most `n` of this wave's memory operations may still be in flight (n <= 16)
__device__ __forceinline__ void ss_wait_outstanding(int n) {
  switch (n) {
    case 0: v3_vmcnt<0>(); break;
    case 2: v3_vmcnt<2>(); break;
    case 4: v3_vmcnt<4>(); break;
    case 6: v3_vmcnt<6>(); break;
    case 8: v3_vmcnt<8>(); break;
    case 12: v3_vmcnt<12>(); break;
    case 16: v3_vmcnt<16>(); break;
    default: v3_vmcnt<0>(); break;
  }
}

// MT: 32-row batch halves (1: M <= 32, 2: M <= 64); R: row groups per work item (8 / R k-quarters per row group);
// P: pieces (1 KB of packed weights = 32 rows x 64 k) per consumer wave at most; NKB > 0: every k-slice is exactly NKB 64-k blocks
// (NKB even, so slices start on a chunk boundary: the headline's q|k|v and o) -- every count below is then a compile-time
// constant: no run-time wait counts, pieces past the slice only in the last round, and the scalar prologue in front of the first
// weight request shrinks to the work-item decode.  NKB = 0: slice lengths from the arguments.
template <int MT, int R, int P, int NKB>
__global__ __launch_bounds__(SS_THREADS) void wss_kernel(SS_ARGS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int KQ = SS_CONSUMERS / R;
  constexpr bool EXACT = NKB > 0;
  constexpr int PH = P / 2;          // pieces per activation stage
  constexpr int XT = MT * 32 * 256;  // one chunk's activation tile: [MT * 32 rows][16 x 16 B], slot j of row r at j ^ (r & 15)
  constexpr int SNC = NKB / 2;                                              // static: chunks of a slice ...
  constexpr int SNC0 = (PH * KQ + 1) / 2 < SNC ? (PH * KQ + 1) / 2 : SNC;   // ... of which stage 0 delivers these
  static_assert(P % 2 == 0 && NKB % 2 == 0 && (!EXACT || (P * KQ >= NKB && (P - 1) * KQ < NKB)), "piece bound");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  SS_TL(0)
  SS_TLC(14)
  const int item = ((int)blockIdx.x & 7) * p_cap + ((int)blockIdx.x >> 3);
  if (item >= p_total) return;
  const int s = p_rb_magic ? (int)__umulhi((uint32_t)item, p_rb_magic) : item, rb = item - s * p_RB;
  const int kb_lo = EXACT ? s * NKB : s * p_kb_base + (s < p_kb_rem ? s : p_kb_rem);
  const int nkb = EXACT ? NKB : p_kb_base + (s < p_kb_rem ? 1 : 0);
  const int c_lo = kb_lo >> 1;
  const int nC = EXACT ? SNC : ((kb_lo + nkb + 1) >> 1) - c_lo;  // activation chunks the slice touches
  // stage 0 (barrier A) delivers the chunks of the first PH pieces of every consumer: 64-k blocks [kb_lo, kb_lo + PH * KQ)
  int nC0 = EXACT ? SNC0 : ((kb_lo + PH * KQ - 1) >> 1) - c_lo + 1;
  if (!EXACT && nC0 > nC) nC0 = nC;

  if (wv >= SS_CONSUMERS) {
    // ============================== activation loaders ============================== //
    constexpr int PPL = MT * 8 / SS_LOADERS;  // 1-KB pieces (4 rows x 256 B) of a chunk tile per loader wave
    const int L = wv - SS_CONSUMERS;
    uint32_t voff[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      int row = (L * PPL + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (row & 15);
      if (row >= p_m) row = p_m - 1;  // rows >= M feed only unstored outputs
      voff[j] = (uint32_t)(row * p_x_stride * 2 + slot * 16);
    }
    const char* xb = (const char*)p_x + (size_t)c_lo * 256;
    if constexpr (EXACT) {
#pragma unroll
      for (int c = 0; c < SNC; ++c)
#pragma unroll
        for (int j = 0; j < PPL; ++j) v3_dma16<false>((uint32_t)(c * XT + (L * PPL + j) * 1024), xb + (size_t)c * 256, voff[j]);
    } else {
      for (int c = 0; c < nC; ++c) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) v3_dma16<false>((uint32_t)(c * XT + (L * PPL + j) * 1024), xb + (size_t)c * 256, voff[j]);
      }
    }
    SS_TL(1)
    if constexpr (EXACT) v3_vmcnt<((SNC - SNC0) * PPL <= 63 ? (SNC - SNC0) * PPL : 0)>();
    else ss_wait_outstanding((nC - nC0) * PPL);
    SS_TL(2)
    ss_barrier();  // A: chunks [0, nC0) of the slice are in LDS
    v3_vmcnt<0>();
    SS_TL(3)
    ss_barrier();  // B: all of it
    return;
  }

  // ================================= consumers ================================= //
  const int r = wv % R, q = wv / R;
  const int rg = rb * R + r;             // row group of 32 weight rows
  const int nl = lane & 31, h = lane >> 5;
  u32x4 w[P];
  u32x2 sc[P];
  {
    // piece (row group rg, 64-k block kb) of the packed layout = 1 KB at ((tile * chunks + kb / 2) * 8 + (kb & 1) * 4 + rg % 4) KB
    // = row-group base + kb * 4 KB: a scalar base per piece + one 32-bit lane offset
    const char* wrow = p_wp + ((size_t)(uint32_t)((rg >> 2) * p_chunks) * 8 + (uint32_t)(rg & 3)) * 1024;
    const char* srow = p_sp + (size_t)(uint32_t)rg * 256;
    const uint32_t wl = (uint32_t)lane * 16, sl = (uint32_t)nl * 8;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int kk = i * KQ + q;
      const bool valid = (EXACT && i * KQ + KQ - 1 < NKB) || kk < nkb;  // (static for every round but the last)
      const int kb = kb_lo + (valid ? kk : 0);  // pieces past the slice re-read its first one (multiplied by zero)
      w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)(uint32_t)kb * 4096 + wl));
      sc[i] = *reinterpret_cast<const u32x2*>(srow + (size_t)(uint32_t)((kb >> 1) >> p_gshift) * (uint32_t)p_n * 8 + sl);
    }
  }
  SS_TL(1)
  int x_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x_off[j] = nl * 256 + (((h * 4 + j) ^ (nl & 15)) * 16);
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

  // Branch-free: a piece past the slice is multiplied with its (s, -z s) pair zeroed, i.e. by exact zeros.  The activation
  // fragments of a piece are read one piece ahead of its MFMAs (inside a stage); every count is static.
  auto read_x = [&](f16x8 (&a)[4][MT], int i) {
    const int kk = i * KQ + q;
    const bool valid = (EXACT && i * KQ + KQ - 1 < NKB) || kk < nkb;
    const int kb = kb_lo + (valid ? kk : 0);  // (a piece past the slice: finite, landed data x exact zeros)
    const unsigned char* xb = lds + ((kb >> 1) - c_lo) * XT;
    const int kx = (kb & 1) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[j][mt] = *reinterpret_cast<const f16x8*>(xb + (x_off[j] ^ kx) + mt * 32 * 256);
  };
  auto mul = [&](const f16x8 (&a)[4][MT], int i) {
    uint32_t s0 = sc[i].x, s1 = sc[i].y;
    if (!(EXACT && i * KQ + KQ - 1 < NKB)) {  // (compile-time false for a static slice's full rounds)
      const uint32_t keep = (i * KQ + q) < nkb ? 0xffffffffu : 0u;
      s0 &= keep;
      s1 &= keep;
    }
    SS_TL_PIECE_LANDED(i, w[i].x, s0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t word = j == 0 ? w[i].x : j == 1 ? w[i].y : j == 2 ? w[i].z : w[i].w;
      const f16x8 wfrag = v3_dequant(word, s0, s1, magic);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, a[j][mt], acc[mt], 0, 0, 0);
    }
    SS_TL_PIECE_DONE(i, acc[MT - 1][15])
  };
  f16x8 a0[4][MT], a1[4][MT];
  ss_barrier();  // A
  SS_TL(2)
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    if (st == 1) ss_barrier();  // B
    read_x(a0, st * PH);
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      if (i + 1 < PH) read_x((i & 1) ? a0 : a1, st * PH + i + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads up here (hipcc sinks them to their use otherwise)
      mul((i & 1) ? a1 : a0, st * PH + i);
    }
  }
  SS_TL(11)

  // The KQ partial sums of a row group meet through the LDS the activation slice no longer needs: [wave][batch row][8 x 16 B],
  // 16-byte slot j of batch row m stored at j ^ ((m >> 1) & 7) (the MFMA-layout writes and the line-layout reads are both
  // conflict-free), summed in wave order, stored as whole 128-byte lines of the plane.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every consumer has read its last activation fragment
  {
    unsigned char* mine = lds + (size_t)wv * (MT * 32 * 128);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 32 + nl;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int quad = 2 * g + h;
        *reinterpret_cast<f32x4*>(mine + m * 128 + ((quad ^ ((m >> 1) & 7)) * 16)) =
            f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  SS_TL(12)
  {
    constexpr int QW = MT * 256 / KQ;  // 16-byte quads of the row group's [MT * 32][32] fp32 output this wave finishes
    float* plane = p_out + (size_t)s * p_m * p_n + (size_t)rg * 32;
#pragma unroll
    for (int f0 = 0; f0 < QW; f0 += 64) {
      const int f = f0 + lane;
      if (QW >= 64 || f < QW) {
        const int F = q * QW + f;
        const int m = F >> 3, quad = F & 7;
        const unsigned char* src = lds + (size_t)(r * (MT * 32 * 128)) + m * 128 + ((quad ^ ((m >> 1) & 7)) * 16);
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int q2 = 1; q2 < KQ; ++q2) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(src + (size_t)q2 * R * (MT * 32 * 128));
          v += o;
        }
        if (m < p_m) *reinterpret_cast<f32x4*>(plane + (size_t)m * p_n + quad * 4) = v;
      }
    }
  }
  SS_TL(13)
  SS_TLC(15)
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct SSPlan {
  int ok = 0;
  int R = 0, S = 0, P = 0, kb_base = 0, kb_rem = 0, RB = 0, total = 0, cap = 0, grid = 0, lds = 0, max_chunks = 0;
};

struct SSKnobs {
  int on = 1;          // LL_GEMM_SS=0: off (A/B against the unit loop)
  int force_r = 0, force_s = 0;  // LL_GEMM_SS_R / LL_GEMM_SS_S: tuning
  SSKnobs() {
    if (const char* e = getenv("LL_GEMM_SS")) on = atoi(e);
    if (const char* e = getenv("LL_GEMM_SS_R")) force_r = atoi(e);
    if (const char* e = getenv("LL_GEMM_SS_S")) force_s = atoi(e);
  }
};
static const SSKnobs& ss_knobs() {
  static const SSKnobs k;
  return k;
}

static int ss_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// the slice geometry of (kbs k-blocks, S slices): most activation chunks any slice touches
static int ss_max_chunks(int kbs, int S) {
  const int base = kbs / S, rem = kbs % S;
  int worst = 0;
  for (int s = 0; s < S; ++s) {
    const int lo = s * base + (s < rem ? s : rem), n = base + (s < rem ? 1 : 0);
    const int nc = ((lo + n + 1) >> 1) - (lo >> 1);
    if (nc > worst) worst = nc;
  }
  return worst;
}

// The plan: (R row groups per item, S k-slices) that minimises a three-term estimate -- a CU's share of the weight stream at the
// chip's HBM rate, its k-slice of the activation matrix at the L2 -> CU rate (the two add: one path into the CU, DESIGN_NOTEBOOK.md 4.2),
// and the S planes the consumer launch reads back.  One workgroup per CU, one round of workgroups.
static SSPlan ss_plan(int64_t m, int64_t n, int64_t k, int group_size) {
  SSPlan best;
  const SSKnobs& kn = ss_knobs();
  if (!kn.on || !ll_w4a16_prepacked_supported(m, n, k, group_size)) return best;
  const int cus = ss_num_cus();
  const int rgs = (int)(n / 32), kbs = (int)(k / 64);
  const int mt = m > 32 ? 2 : 1;
  double best_cost = 1e30;
  for (int R = 1; R <= 8; R *= 2) {
    if (rgs % R) continue;
    if (kn.force_r && R != kn.force_r) continue;
    const int KQ = 8 / R, RB = rgs / R;
    for (int S = 1; S <= SS_MAX_SLICES; ++S) {
      if (kn.force_s && S != kn.force_s) continue;
      if (S > kbs) break;
      const int total = RB * S;
      if (total > cus) continue;
      const int nkb = (kbs + S - 1) / S;
      const int pw = (nkb + KQ - 1) / KQ;
      if (pw > 8) continue;
      const int mc = ss_max_chunks(kbs, S);
      if (mc > SS_MAX_CHUNKS) continue;
      // Measured on MI355X (benchmarks/gemm_short.py, DESIGN_NOTEBOOK.md 4.6): a CU's share of the weight stream arrives at the chip's
      // HBM rate (~27 KB / us / CU), its activation slice at ~100 KB / us, its planes leave at ~25 KB / us (the dirty bytes are
      // written back at the kernel boundary), and every plane costs the consumer launch ~0.15 us.
      const int slots = ((pw + 1) / 2) * 2 * KQ;                        // pieces multiplied per row group (incl. zeroed ones)
      const double w_us = (double)R * nkb * 1.0625 / 27.0;
      const double hbm = (double)n * (double)k * 0.5625 / 6.5e6;       // the whole stream at ~6.5 TB/s, us
      const double x_us = (double)mc * mt * 8.0 / 100.0;
      const double p_us = (double)R * mt * 4.0 / 25.0;
      const double cost = (hbm > w_us ? hbm : w_us) + x_us + p_us + 0.15 * S + 0.02 * R * (slots - nkb);
      if (cost < best_cost) {
        best_cost = cost;
        best.ok = 1; best.R = R; best.S = S; best.P = ((pw + 1) / 2) * 2;
        best.kb_base = kbs / S; best.kb_rem = kbs % S; best.RB = RB; best.total = total;
        best.cap = (total + 7) / 8; best.grid = best.cap * 8; best.max_chunks = mc;
        const int xbytes = mc * mt * 32 * 256, rbytes = 8 * mt * 32 * 128;
        best.lds = xbytes > rbytes ? xbytes : rbytes;
      }
    }
  }
  return best;
}

// planes the short-stream engine leaves for (m, n, k); 0: the launch is not its (gemm_w4_v3.hip asks)
int ss_partials_slices(int64_t m, int64_t n, int64_t k, int group_size) {
  const SSPlan pl = ss_plan(m, n, k, group_size);
  return pl.ok ? pl.S : 0;
}

// Host-side introspection (tests, DESIGN_NOTEBOOK.md; no device work): [0] 1 if the engine takes the split-K partial launch, [1] grid,
// [2] R, [3] S, [4] pieces per wave (template bound), [5] k-blocks per slice (base), [6] slices with one more, [7] LDS bytes.
extern "C" int ll_w4a16_short_plan(int64_t m, int64_t n, int64_t k, int group_size, int32_t* out8) {
  if (!out8) return LL_ERR_ARG;
  const SSPlan pl = ss_plan(m, n, k, group_size);
  const int v[8] = {pl.ok, pl.grid, pl.R, pl.S, pl.P, pl.kb_base, pl.kb_rem, pl.lds};
  for (int i = 0; i < 8; ++i) out8[i] = v[i];
  return LL_OK;
}

template <int MT, int R, int P, int NKB>
static void ss_go(const SSParams& p, const SSPlan& pl, hipStream_t st) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wss_kernel<MT, R, P, NKB>, hipFuncAttributeMaxDynamicSharedMemorySize, SS_MAX_CHUNKS * 2 * 32 * 256);
    attr_set[dev] = true;
  }
  wss_kernel<MT, R, P, NKB><<<dim3((unsigned)pl.grid), SS_THREADS, pl.lds, st>>>(SS_PASS(p));
}

// Shapes / pointers were validated by the caller (v3_launch).
int ss_launch(void* out, const void* x, const void* wpacked, const void* spacked, int64_t m, int64_t n, int64_t k, int group_size,
              int64_t x_stride_m, void* stream) {
  const SSPlan pl = ss_plan(m, n, k, group_size);
  if (!pl.ok) return LL_ERR_SHAPE;
  SSParams p{};
  p.out = (float*)out; p.x = (const uint16_t*)x; p.wp = (const char*)wpacked; p.sp = (const char*)spacked;
  p.m = (int)m; p.n = (int)n; p.x_stride = (int)x_stride_m;
  p.chunks = (int)(k / 128);
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  p.kb_base = pl.kb_base; p.kb_rem = pl.kb_rem; p.RB = pl.RB; p.total = pl.total; p.cap = pl.cap;
  p.rb_magic = pl.RB > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)pl.RB - 1) / (uint64_t)pl.RB) : 0u;  // exact for item * RB < 2^32
  p.tl = nullptr;
  SS_DEBUG_SET(p)
  hipStream_t st = (hipStream_t)stream;
  const bool two = m > 32;
  // static slice lengths for the headline's two launches (q|k|v: R 4 x 8 blocks; o: R 2 x 14 blocks); everything else reads them
  // from the arguments
  const int nkb_static = pl.kb_rem == 0 && ((pl.R == 4 && pl.kb_base == 8 && pl.P == 4) || (pl.R == 2 && pl.kb_base == 14 && pl.P == 4))
                             ? pl.kb_base : 0;
#define SS_GO(MTT, RR, PP) ss_go<MTT, RR, PP, 0>(p, pl, st)
#define SS_CASE(RR)                                                                  \
  case RR:                                                                           \
    if (two) { if (pl.P == 2) SS_GO(2, RR, 2); else if (pl.P == 4) SS_GO(2, RR, 4); else if (pl.P == 6) SS_GO(2, RR, 6); else SS_GO(2, RR, 8); } \
    else { if (pl.P == 2) SS_GO(1, RR, 2); else if (pl.P == 4) SS_GO(1, RR, 4); else if (pl.P == 6) SS_GO(1, RR, 6); else SS_GO(1, RR, 8); }     \
    break;
  if (nkb_static == 8) {
    if (two) ss_go<2, 4, 4, 8>(p, pl, st); else ss_go<1, 4, 4, 8>(p, pl, st);
    return LL_LAUNCH_CHECK();
  }
  if (nkb_static == 14) {
    if (two) ss_go<2, 2, 4, 14>(p, pl, st); else ss_go<1, 2, 4, 14>(p, pl, st);
    return LL_LAUNCH_CHECK();
  }
  switch (pl.R) {
    SS_CASE(1)
    SS_CASE(2)
    SS_CASE(4)
    SS_CASE(8)
    default: return LL_ERR_SHAPE;
  }
#undef SS_CASE
#undef SS_GO
  return LL_LAUNCH_CHECK();
}
