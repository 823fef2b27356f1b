// W4A16 dequant-GEMM, decode engine, fourth generation (round 5): the row-group loop.  Same semantics, same load-time layouts
// (ll_w4a16_pack_weights / ll_w4a16_pack_scales) and the same C entry (ll_w4a16_matmul_prepacked) as gemm_w4_v3.hip;
// reference: lite_llama/kernels/quantization/w4a16.py:28-207, fused epilogue kernels/swiglu.py:45-65.
//
// What changed against the third generation, and why (DESIGN_NOTEBOOK.md 4.5):
//   * DECOMPOSITION.  A workgroup owns NRG consecutive 32-row groups of the weight matrix (4 or 5 for the fused gate|up:
//     1184 groups over 256 CUs) and a contiguous range of 128-k chunks -- ALL of K for the launches that write finished
//     outputs.  No stream-K: no fp32 slabs written and read back (128 KB per CU and launch), no merge counters, no owner
//     tail (~7 us of the 31-us gate|up launch), no unit walkers (the scalar state that spilled).  Split-K partial launches
//     (o, down, q|k|v: epilogue 2) keep their k-slices and planes, only the body changes.
//   * CONSUMERS.  Eight consumer waves = 4 k-QUARTERS of every chunk (two MFMA k-steps each) x 2 halves of the workgroup's row
//     groups (3 + 2 of 5), both batch halves in one wave: every weight word is read from LDS and dequantised exactly once per
//     workgroup, every activation fragment is read twice (v3: four times; 72 KB of operand reads per 128-row unit against
//     43 KB per 160-row unit here).  The accumulators stay in their wave for the whole launch; the four k-quarters meet once,
//     at the end, through the LDS the rings no longer need.  (A first form with FOUR consumer waves, one per SIMD, all row groups
//     each -- no redundancy at all -- measured compute-bound: one wave issues ~200 instructions per unit at ~5 cycles each
//     and nothing covers its dequantisation VALU; DESIGN_NOTEBOOK.md 4.5.)
//   * LOADERS.  Four loader waves (two for weights + scale pairs, two for the activation tile), LDS-DMA only, R-slot rings
//     filled R - 1 units ahead; one s_barrier per unit for the twelve waves.
#include <stdlib.h>

#include "common.h"
#include "gemm_debug.h"  // V4_TL: in-kernel stamps of -DV4_TIMELINE builds (empty in the product build)
#include "gemm_w4_common.h"

#define V4_THREADS 768
#define V4_X_SLOT 16384  // [64 rows][16 x 16 B], slot j of row r stored at j ^ (r & 15) (the v3 image)
// Ring slots per stream (weights + scale pairs / activation tile), by row groups per workgroup.  A stream's units are requested
// R - 1 ahead, of which V4_AHEAD + 1 slots are the ones the consumers may be reading: R - 1 - V4_AHEAD unit periods cover the
// latency of a request (HBM under load: ~2700 cycles, in-kernel stamps) -- the weight stream wants depth, the activation tile
// (L2 hits after the warm-up reads at the start of the launch) less.
#ifndef V4_RW5
#define V4_RW5 7
#endif
#ifndef V4_RX5
#define V4_RX5 4
#endif
#ifndef V4_RW4
#define V4_RW4 8
#endif
#ifndef V4_RX4
#define V4_RX4 5
#endif
#ifndef V4_XWARM
#define V4_XWARM 1  // every workgroup reads its share of the launch's activation lines once at entry: the XCD's L2 is warm from then on
#endif
#ifndef V4_AHEAD
#define V4_AHEAD 2  // after the barrier that ends unit u the consumers may touch units <= u + V4_AHEAD
#endif


template <int NRGT>
struct V4Lds {
  static constexpr int W_BYTES = NRGT * 2048;          // [k-half 2][row group NRGT] x 1 KB: the v3 pieces of the row groups
  static constexpr int W_SLOT = W_BYTES + NRGT * 256;  // + the rows' (s, s | -z*s, -z*s) pairs, 8 B per row
  static constexpr int RW = NRGT >= 5 ? V4_RW5 : V4_RW4, RX = NRGT >= 5 ? V4_RX5 : V4_RX4;
  static_assert(RW >= V4_AHEAD + 2 && RX >= V4_AHEAD + 2, "ring depth");
  static constexpr int OFF_W = 0;
  static constexpr int OFF_X = RW * W_SLOT;
  static constexpr int RING = OFF_X + RX * V4_X_SLOT;
  static constexpr int RED = NRGT * 2 * 4 * 4096;  // end of launch: [fragment (row group, batch half)][k-quarter] x 4 KB
  static constexpr int BYTES = RING > RED ? RING : RED;
  static_assert(BYTES <= 160 * 1024, "LDS");
};

struct V4Params {
  void* out;
  const uint16_t* x;
  const void* wp;  // packed weights  [N/128][K/128][wave 8 = (kh, ng)][lane 64] x 16 B   (ll_w4a16_pack_weights)
  const void* sp;  // packed scales   [K/g][N] x 8 B                                      (ll_w4a16_pack_scales)
  const uint16_t* bias;
  int64_t m, n, x_stride;
  int chunks;  // K / 128
  int gshift;  // log2(group_size / 128)
  int epi;     // 0: out[m, n] fp16;  1: rows are (gate_j, up_j) pairs -> out[m, n/2] = silu(gate) * up;  2: fp32 planes [slice][m][n]
  int ks;      // k-slices: workgroup b = tile * ks + slice (ks | 8: slice == XCD under the round-robin dispatch, so an XCD's
               // L2 only ever sees ONE k-slice of the activation matrix -- the v3 map)
  int rbase, rrem;  // tile t owns row groups [t * rbase + min(t, rrem), + rbase + (t < rrem))
  int cbase, crem;  // slice j owns chunks likewise
  int xw_peers;     // > 0: workgroups per XCD that share a k-slice of the activations (L2 warm-up reads at entry); 0: none
  V4_DEBUG_FIELDS
};

__device__ __forceinline__ void v4_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N> struct V4Int { static constexpr int value = N; };

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the loaders' operation count per unit depends on the workgroup's row
// groups); anything above the table waits for the table's last entry, which is only earlier than necessary
__device__ __forceinline__ void v4_vmcnt_dyn(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
#define V4_C(N) case N: v3_vmcnt<N>(); break;
  switch (n) {
    V4_C(0) V4_C(1) V4_C(2) V4_C(3) V4_C(4) V4_C(5) V4_C(6) V4_C(7) V4_C(8) V4_C(9) V4_C(10) V4_C(11) V4_C(12) V4_C(13)
    V4_C(14) V4_C(15) V4_C(16) V4_C(17) V4_C(18) V4_C(19) V4_C(20) V4_C(21) V4_C(22) V4_C(23) V4_C(24) V4_C(25) V4_C(26)
    V4_C(27) V4_C(28) V4_C(29) V4_C(30) V4_C(31) V4_C(32) V4_C(33) V4_C(34) V4_C(35) V4_C(36) V4_C(37) V4_C(38) V4_C(39)
    V4_C(40) V4_C(41) V4_C(42) V4_C(43) V4_C(44) V4_C(45) V4_C(46) V4_C(47) V4_C(48)
    default: if (n < 0) v3_vmcnt<0>(); else v3_vmcnt<48>(); break;
  }
#undef V4_C
}

template <int NRGT, int MT>
__global__ __launch_bounds__(V4_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgemm4_kernel(const V4Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using LD = V4Lds<NRGT>;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (int)blockIdx.x;
  const int tile = p.ks > 1 ? (int)((unsigned)b / (unsigned)p.ks) : b;
  const int slice = b - tile * p.ks;
  const int rg0 = tile * p.rbase + (tile < p.rrem ? tile : p.rrem);
  const int nrg = p.rbase + (tile < p.rrem ? 1 : 0);
  const int c0 = slice * p.cbase + (slice < p.crem ? slice : p.crem);
  const int cnt = p.cbase + (slice < p.crem ? 1 : 0);
  if (cnt <= 0 || nrg <= 0) return;
  V4_TL(0)
  V4_TL_REAL(62)

  if (wv >= 8) {
    // ======================================== loaders ======================================== //
#if defined(V4_LOADER_PRIO) && V4_LOADER_PRIO > 0  // A/B: the loaders are the youngest waves of their SIMDs (arbitration losers by age)
    __builtin_amdgcn_s_setprio(V4_LOADER_PRIO);
#endif
    const int L = wv & 1;
    int issued = 0;
    const bool wl = wv < 10;  // waves 8, 9: weights + scale pairs; 10, 11: the activation tile
    uint32_t dst = wl ? LD::OFF_W : LD::OFF_X;
    const int R = wl ? LD::RW : LD::RX;
    const uint32_t ring_lo = dst, ring_hi = dst + R * (wl ? LD::W_SLOT : V4_X_SLOT), step = wl ? LD::W_SLOT : V4_X_SLOT;
    int ops;
    // per-lane source offsets
    uint32_t voff[8];
    const char* base;
    const char* sbase = nullptr;
    if (wl) {
      // weight loader L: the k-half-L piece of every row group (1 KB each); L == 1 also the scale pairs
#pragma unroll
      for (int r = 0; r < NRGT; ++r) {
        const int rg = rg0 + (r < nrg ? r : 0);
        voff[r] = (uint32_t)((rg >> 2) * p.chunks) * 8192u + (uint32_t)((L * 4 + (rg & 3)) * 1024 + lane * 16);
      }
      base = (const char*)p.wp + (size_t)c0 * 8192;
      ops = nrg + (L ? (nrg >= 4 ? 1 + (nrg - 4) : nrg) : 0);
    } else {
      // activation loader L: pieces XP * L .. + XP - 1 (4 rows x 256 B each); LDS image row r, 16-B slot j <- source slot j ^ (r & 15)
      constexpr int XP = 4 * MT;
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        int64_t r = (XP * L + j) * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ (int)(r & 15);
        if (r >= p.m) r = p.m - 1;  // rows >= M feed only unstored outputs
        voff[j] = (uint32_t)(r * p.x_stride * 2 + slot * 16);
      }
      base = (const char*)p.x + (size_t)c0 * 256;
      ops = XP;
      if (V4_XWARM && p.xw_peers > 0) {
        // L2 warm-up: the workgroups that share this k-slice of the activation matrix ON THIS XCD (b = 8 i + xcd under the
        // round-robin dispatch) touch 1 / peers of its 128-byte lines each -- one dword per line and lane, results unused.
        // Speed only: nothing depends on where a workgroup really runs.
        const int lines_per_row = 2 * cnt, nlines = (int)p.m * lines_per_row;
        const int i = b >> 3, share = (nlines + p.xw_peers - 1) / p.xw_peers;
        const int lo = i * share, hi = lo + share < nlines ? lo + share : nlines;
        for (int idx = lo + L * 64 + lane; idx < hi; idx += 128) {
          const int r = idx / lines_per_row, l = idx - r * lines_per_row;
          const char* a = (const char*)p.x + (size_t)r * (size_t)p.x_stride * 2 + (size_t)c0 * 256 + (size_t)l * 128;
          uint32_t dummy;
          asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(a) : "memory");
        }
      }
    }
    int c = c0;
    auto issue = [&]() {
      if (wl) {
#pragma unroll
        for (int r = 0; r < NRGT; ++r)
          if (r < nrg) v3_dma16<(V3_NT_W & 1) != 0>(dst + (L * NRGT + r) * 1024, base, voff[r]);
        if (L) {
          sbase = (const char*)p.sp + ((size_t)(c >> p.gshift) * (size_t)p.n + (size_t)rg0 * 32) * 8;
          if (nrg >= 4) {
            v3_dma16<false>(dst + LD::W_BYTES, sbase, (uint32_t)(lane * 16));
#pragma unroll
            for (int r = 4; r < NRGT; ++r)
              if (r < nrg) v3_dma4<false>(dst + LD::W_BYTES + r * 256, sbase, (uint32_t)(r * 256 + lane * 4));
          } else {
#pragma unroll
            for (int r = 0; r < (NRGT < 4 ? NRGT : 3); ++r)
              if (r < nrg) v3_dma4<false>(dst + LD::W_BYTES + r * 256, sbase, (uint32_t)(r * 256 + lane * 4));
          }
        }
        base += 8192;
      } else {
        constexpr int XP = 4 * MT;
#pragma unroll
        for (int j = 0; j < XP; ++j)
          v3_dma16<false>(dst + (XP * L + j) * 1024, base, voff[j]);
        base += 256;
      }
      ++c;
      ++issued;
      dst = dst + step == ring_hi ? ring_lo : dst + step;
    };
    // Prologue: only what the consumers' first step needs is requested before P0 (a wave's operations return in order: unit 0
    // lands first); the ring is filled two units per step afterwards.
    const int pre = cnt < V4_AHEAD ? cnt : V4_AHEAD;
    for (int i = 0; i < pre; ++i) issue();
    v3_vmcnt<0>();
    v4_barrier();  // P0
    V4_TL(1)
    for (int u = 0; u < cnt; ++u) {
      const int want = cnt < u + R ? cnt : u + R;  // the slot of unit u - 1 is free since the barrier that ended it
      if (issued < want) issue();
      if (issued < want) issue();
      if (u < 14) { V4_TL(2 + 4 * u) }
      const int need = cnt < u + 1 + V4_AHEAD ? cnt : u + 1 + V4_AHEAD;
      v4_vmcnt_dyn((issued - need) * ops);
      if (u < 14) { V4_TL(3 + 4 * u) }
      v4_barrier();  // B_u
      if (u < 14) { V4_TL(4 + 4 * u) }
    }
    v4_barrier();  // the consumers' k-quarter exchange
    V4_TL(61)
    return;
  }

  // ======================================= consumers ======================================= //
  // wave = (k-quarter q, row-group half par): k-half kh, MFMA k-steps 2 jp, 2 jp + 1 of the v3 lane layout; row groups
  // par * NA .. + NA - 1 of the workgroup's.  Waves q and q + 4 share a SIMD (round-robin placement): 3 + 2 row groups = the
  // SIMD's 20 MFMAs per unit, the VALU stream of one covering the MFMAs of the other.
  constexpr int NA = (NRGT + 1) / 2;
  const int q = wv & 3, par = wv >> 2;
  const int kh = q >> 1, jp = q & 1;
  const int nl = lane & 31, h = lane >> 5;
  const int w_off = (kh * NRGT + par * NA) * 1024 + lane * 16 + jp * 8;
  const int s_off = LD::W_BYTES + par * NA * 256 + nl * 8;
  const int nmine = nrg - par * NA < NA ? nrg - par * NA : NA;  // this wave's row groups that exist (>= 0)
  int x_off[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) x_off[jj] = nl * 256 + (((kh * 8 + h * 4 + 2 * jp + jj) ^ (nl & 15)) * 16);

  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  float* red = reinterpret_cast<float*>(lds);

  // The unit loop of ONE consumer wave over NR row groups (a compile-time count: NA for the first half's waves, NRGT - NA for the
  // second's -- two instances behind one wave-uniform branch).  The body is BRANCH-FREE: a workgroup's last row group may not
  // exist (4 of 5); its wave multiplies whatever its never-written LDS block holds and the result is dropped at the end -- so the
  // whole unit is one scheduling region, and the dequantisation of item i + 1 (14 VALU) is interleaved under the MT MFMAs of item i
  // (`sched_group_barrier`).  Measured against the first form of this file (a branch per row group, each group's VALU block in front
  // of its MFMAs): +-0.3 us per launch -- the compute-only build stays at 19.8 us, because its cost is ADDITIVE in three parts that
  // interleaving inside a wave does not overlap: operand reads + barriers alone 13 us, + MFMAs 3.7, + dequantisation 3.2 (DESIGN_NOTEBOOK.md 4.5).
  auto consume = [&](auto nr_tag) {
    constexpr int NR = decltype(nr_tag)::value;
    constexpr int NI = 2 * NR;  // items of a unit: (k-step jj, row group r)
    f32x16 acc[NR][MT];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][mt][e] = 0.f;
    struct Ops {
      u32x2 w[NR];
      u32x2 s[NR];
    };
    auto read_ws = [&](Ops& o, int wslot) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        o.w[r] = *reinterpret_cast<const u32x2*>(lds + wslot + w_off + r * 1024);
        o.s[r] = *reinterpret_cast<const u32x2*>(lds + wslot + s_off + r * 256);
      }
    };
    auto read_x = [&](f16x8 (&xf)[MT], int xslot, int jj) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const f16x8*>(lds + xslot + x_off[jj] + mt * 32 * 256);
    };
    auto dq = [&](const Ops& o, int i) -> f16x8 {  // item i = jj * NR + r
      const int jj = i / NR, r = i % NR;
      const uint32_t word = jj == 0 ? o.w[r].x : o.w[r].y;
      return v3_dequant(word, o.s[r].x, o.s[r].y, magic);
    };
    int ws_cur = LD::OFF_W, xs_cur = LD::OFF_X;
    Ops opA, opB;
    f16x8 x0[MT], x1[MT];
    v4_barrier();  // P0: units 0 and 1 have landed
    V4_TL(1)
    read_ws(opA, ws_cur);
    read_x(x0, xs_cur, 0);
    int u = 0;
    // One unit: the second k-step's fragments and the NEXT unit's weight words / scale pairs / first fragments are read while
    // this unit is multiplied (they landed before the previous barrier and stay in flight across this one).
#define V4_STEP(CUR, NXT)                                                                                   \
    {                                                                                                       \
      const bool more = u + 1 < cnt;                                                                        \
      const int ws_n = !more ? ws_cur : (ws_cur + LD::W_SLOT == LD::OFF_W + LD::RW * LD::W_SLOT ? LD::OFF_W : ws_cur + LD::W_SLOT); \
      const int xs_n = !more ? xs_cur : (xs_cur + V4_X_SLOT == LD::OFF_X + LD::RX * V4_X_SLOT ? LD::OFF_X : xs_cur + V4_X_SLOT);    \
      read_x(x1, xs_cur, 1);                                                                                \
      read_ws(NXT, ws_n);                                                                                   \
      f16x8 wf[2];                                                                                          \
      wf[0] = dq(CUR, 0);                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                      \
        if (i + 1 < NI) wf[(i + 1) & 1] = dq(CUR, i + 1);                                                   \
        if (i == NR) read_x(x0, xs_n, 0); /* the next unit's first fragments: k-step 0's last use of x0 was item NR - 1 */   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
          acc[i % NR][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i & 1], i < NR ? x0[mt] : x1[mt], acc[i % NR][mt], 0, 0, 0); \
        if (i + 1 < NI) {                                                                                   \
          _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA of item i */                \
            __builtin_amdgcn_sched_group_barrier(0x002, 14 / MT, 0); /* its share of item i + 1's dequantisation */ \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      v4_barrier();                                                                                         \
      if (u < 14) { V4_TL(4 + 4 * u) }                                                                      \
      ws_cur = ws_n;                                                                                        \
      xs_cur = xs_n;                                                                                        \
      ++u;                                                                                                  \
    }
    for (;;) {
      V4_STEP(opA, opB)
      if (u >= cnt) break;
      V4_STEP(opB, opA)
      if (u >= cnt) break;
    }
#undef V4_STEP
    // ---- the four k-quarters meet: every wave parks the fragments of its EXISTING row groups ----
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r < nmine) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float* dstf = red + (((par * NA + r) * MT + mt) * 4 + q) * 1024;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dstf + (g * 64 + lane) * 4) =
                f32x4{acc[r][mt][4 * g], acc[r][mt][4 * g + 1], acc[r][mt][4 * g + 2], acc[r][mt][4 * g + 3]};
        }
      }
    }
  };
  if (par == 0) consume(V4Int<NA>{});
  else consume(V4Int<(NRGT - NA > 0 ? NRGT - NA : 1)>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  v4_barrier();  // ... then the eight waves finish fragments w, w + 8, ...
  V4_TL(60)

  auto swap32 = [](uint32_t& a, uint32_t& bb) {  // lanes h = 1 of `a` <-> lanes h = 0 of `bb`
    const auto rr = __builtin_amdgcn_permlane32_swap(a, bb, false, false);
    a = rr[0];
    bb = rr[1];
  };
  const bool has_bias = p.bias != nullptr;
  const int nfr = nrg * MT;
  for (int f = wv; f < nfr; f += 8) {
    const int r = MT == 2 ? f >> 1 : f, mt = MT == 2 ? (f & 1) : 0;
    float v[16];
    {
      const float* src = red + (f * 4) * 1024 + lane * 4;
      f32x4 t[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) t[s][g] = *reinterpret_cast<const f32x4*>(src + s * 1024 + g * 256);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] = ((t[0][g][e] + t[1][g][e]) + t[2][g][e]) + t[3][g][e];  // fixed order
    }
    const int64_t mrow = nl + mt * 32;
    const bool row_ok = mrow < p.m;
    const int64_t ncol = (int64_t)(rg0 + r) * 32;  // first weight row (= output column) of the row group
    if (p.epi == 2) {
      // split-K partial mode: the fp32 plane of this k-slice, summed by the consumer kernel (ll_skip_rmsnorm_partials, the
      // decode attention) in slice order
      if (row_ok) {
        float* dstp = reinterpret_cast<float*>(p.out) + ((int64_t)slice * p.m + mrow) * p.n + ncol + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dstp + 8 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      }
      continue;
    }
    // a lane holds, for its batch row, the weight rows 8g + 4h .. + 3 of the group's 32 (g = 0..3): the v3 epilogue
    if (has_bias) {
      uint2 bb[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[g] = *reinterpret_cast<const uint2*>(p.bias + ncol + 8 * g + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[4 * g + 0] += f16_bits_to_f32((uint16_t)(bb[g].x & 0xffffu));
        v[4 * g + 1] += f16_bits_to_f32((uint16_t)(bb[g].x >> 16));
        v[4 * g + 2] += f16_bits_to_f32((uint16_t)(bb[g].y & 0xffffu));
        v[4 * g + 3] += f16_bits_to_f32((uint16_t)(bb[g].y >> 16));
      }
    }
    uint32_t lo[4], hi[4], sw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint16_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f32_to_f16_bits(v[4 * g + e]);
      lo[g] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
      hi[g] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
      if (p.epi) {
        // weight rows 2j / 2j+1 are gate_j / up_j: both land in this lane.  Same arithmetic as the stand-alone kernels: the two
        // GEMM outputs rounded to fp16, then silu(g) * u in fp32 (common.h::ll_sigmoidf, the one definition)
        const float g0 = f16_bits_to_f32(o[0]), u0 = f16_bits_to_f32(o[1]);
        const float g1 = f16_bits_to_f32(o[2]), u1 = f16_bits_to_f32(o[3]);
        const uint32_t s0 = f32_to_f16_bits(ll_silu_mul_f32(g0, u0));
        const uint32_t s1 = f32_to_f16_bits(ll_silu_mul_f32(g1, u1));
        sw[g] = s0 | (s1 << 16);
      }
    }
    // lanes nl and nl + 32 hold the interleaving pieces of the SAME batch row: one v_permlane32_swap per register and every
    // lane writes 16 contiguous bytes per store
    const int64_t n0 = ncol + 16 * h;
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
    if (p.epi) {
      swap32(sw[0], sw[2]);
      swap32(sw[1], sw[3]);
      if (row_ok) *reinterpret_cast<u32x4*>(outp + mrow * (p.n >> 1) + (n0 >> 1)) = u32x4{sw[0], sw[2], sw[1], sw[3]};
    } else {
      swap32(lo[0], lo[2]);
      swap32(hi[0], hi[2]);
      swap32(lo[1], lo[3]);
      swap32(hi[1], hi[3]);
      if (row_ok) {
        *reinterpret_cast<u32x4*>(outp + mrow * p.n + n0) = u32x4{lo[0], hi[0], lo[2], hi[2]};
        *reinterpret_cast<u32x4*>(outp + mrow * p.n + n0 + 8) = u32x4{lo[1], hi[1], lo[3], hi[3]};
      }
    }
  }
  V4_TL(61)
  V4_TL_REAL(63)
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct V4Knobs {
  int on = 1;       // LL_GEMM4: bit 0 = launches that write finished outputs (epilogues 0 / 1), bit 1 = split-K partial launches
                    // (default 1: same box, the fused gate|up 32.1 -> 28.3 us; the 3.5 - 18-unit partial launches are fixed cost and
                    // measure +-0 / +0.5 us on this body, DESIGN_NOTEBOOK.md 4.5)
  int rg_part = 4;  // LL_GEMM4_RGP: row groups per tile of a split-K partial launch (2 or 4)
  int min_fill = 3; // finished-output launches take this engine from min_fill row groups per CU on
  int small = 0;    // LL_GEMM4_SMALL: ... and, below that fill, from this many row groups on (0: never; round 6, TP shards)
  V4Knobs() {
    if (const char* e = getenv("LL_GEMM4")) on = atoi(e);
    if (const char* e = getenv("LL_GEMM4_RGP")) rg_part = atoi(e) == 2 ? 2 : 4;
    if (const char* e = getenv("LL_GEMM4_MINFILL")) min_fill = atoi(e);
    if (const char* e = getenv("LL_GEMM4_SMALL")) small = atoi(e);
  }
};
static const V4Knobs& v4_knobs() {
  static const V4Knobs k;
  return k;
}

static int v4_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// row groups per workgroup of a finished-output launch; 0: not served here
static int v4_full_nrgt(int64_t n) {
  const int cus = v4_num_cus();
  const int64_t rgs = n / 32;
  const int per = (int)((rgs + cus - 1) / cus);
  if (rgs < (int64_t)v4_knobs().min_fill * cus) {
    // fewer row groups than min_fill per CU -- the fused gate|up of a TP shard (592 / 296 / 148 row groups at TP 2 / 4 / 8): one
    // workgroup per CU (or per row group) with 1 .. 4 row groups each on the 2- / 4-group instances, from `small` row groups on
    const int small = v4_knobs().small;
    if (small <= 0 || rgs < small) return 0;
    return per <= 2 ? 2 : per <= 4 ? 4 : 0;
  }
  return (per == 2 || per == 4 || per == 5) ? per : 0;
}

int v4_wants(int64_t m, int64_t n, int64_t k, int group_size, int epilogue) {
  (void)m; (void)k; (void)group_size;
  if ((epilogue >> 8) & 3) return 0;  // a forced v3 tile width (tests / tuning)
  const int epi = epilogue & 3;
  const V4Knobs& kn = v4_knobs();
  if (epi == 2) return (kn.on & 2) ? 1 : 0;
  return (kn.on & 1) && v4_full_nrgt(n) ? 1 : 0;
}

// Host-side introspection (tests, DESIGN_NOTEBOOK.md; no device work): the row-group engine's plan for a finished-output launch
// (epilogue 0 / 1) of n weight rows as 8 ints -- [0] 1 if the engine takes the launch (else the unit loop does), [1] grid, [2] row
// groups per workgroup (template bound), [3] rbase, [4] rrem (workgroup t owns groups [t * rbase + min(t, rrem), + rbase + (t < rrem))),
// [5] compute units assumed, [6] LDS bytes, [7] 0.
extern "C" int ll_w4a16_v4_plan(int64_t m, int64_t n, int64_t k, int group_size, int epilogue, int32_t* out8) {
  if (!out8) return LL_ERR_ARG;
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size)) return LL_ERR_SHAPE;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  const int nrgt = ((epilogue & 3) == 2) ? 0 : v4_full_nrgt(n);
  out8[5] = v4_num_cus();
  if (!nrgt || !v4_wants(m, n, k, group_size, epilogue)) return LL_OK;
  const int rgs = (int)(n / 32);
  int ntiles = v4_num_cus();
  if (ntiles > rgs) ntiles = rgs;
  out8[0] = 1; out8[1] = ntiles; out8[2] = nrgt; out8[3] = rgs / ntiles; out8[4] = rgs % ntiles;
  out8[6] = nrgt == 2 ? V4Lds<2>::BYTES : nrgt == 4 ? V4Lds<4>::BYTES : V4Lds<5>::BYTES;
  return LL_OK;
}

template <int NRGT, int MT>
static void v4_go(const V4Params& p, int grid, hipStream_t st) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wgemm4_kernel<NRGT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, V4Lds<NRGT>::BYTES);
    attr_set[dev] = true;
  }
  wgemm4_kernel<NRGT, MT><<<dim3((unsigned)grid), V4_THREADS, V4Lds<NRGT>::BYTES, st>>>(p);
}

// Shapes were validated by the caller (v3_launch); kslices = the k-split of the partial mode (= ll_w4a16_partials_count).
int v4_launch(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias, int64_t m, int64_t n, int64_t k,
              int group_size, int64_t x_stride_m, int epilogue, int kslices, void* stream) {
  V4Params p{};
  p.out = out; p.x = (const uint16_t*)x; p.wp = wpacked; p.sp = spacked; p.bias = (const uint16_t*)bias;
  p.m = m; p.n = n; p.x_stride = x_stride_m;
  p.chunks = (int)(k / 128);
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  p.epi = epilogue & 3;
  const int rgs = (int)(n / 32);
  int nrgt, ntiles;
  if (p.epi == 2) {
    nrgt = v4_knobs().rg_part;
    ntiles = rgs / nrgt;  // n % 128 == 0
    p.rbase = nrgt; p.rrem = 0;
    p.ks = kslices;
  } else {
    nrgt = v4_full_nrgt(n);
    if (!nrgt) return LL_ERR_SHAPE;
    ntiles = v4_num_cus();
    if (ntiles > rgs) ntiles = rgs;
    p.rbase = rgs / ntiles; p.rrem = rgs % ntiles;
    p.ks = 1;
  }
  if (p.ks < 1 || p.ks > p.chunks) return LL_ERR_SHAPE;
  p.cbase = p.chunks / p.ks; p.crem = p.chunks % p.ks;
  V4_DEBUG_SET(p)
  const int grid = ntiles * p.ks;
  p.xw_peers = (p.ks == 1 || p.ks == 2 || p.ks == 4 || p.ks == 8) && grid % 8 == 0 ? grid / 8 : 0;
  hipStream_t st = (hipStream_t)stream;
  const bool two = m > 32;
  switch (nrgt) {
    case 2: if (two) v4_go<2, 2>(p, grid, st); else v4_go<2, 1>(p, grid, st); break;
    case 4: if (two) v4_go<4, 2>(p, grid, st); else v4_go<4, 1>(p, grid, st); break;
    case 5: if (two) v4_go<5, 2>(p, grid, st); else v4_go<5, 1>(p, grid, st); break;
    default: return LL_ERR_SHAPE;
  }
  return LL_LAUNCH_CHECK();
}
