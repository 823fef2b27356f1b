// W4A16 dequant-GEMM, decode engine, third generation (M <= 64, group_size = 128 * 2^j) over weights
// PRE-PACKED at load time (ll_w4a16_pack_weights) -- the load-time layout slot the reference reserves in
// lite_llama/models/quantization/_layout/__init__.py:1-6.  Semantics: lite_llama/kernels/quantization/w4a16.py:28-207
// (out[m, n] = sum_k x[m, k] * (nib(n, k) - z[n, k/g]) * s[n, k/g] (+ bias), fp32 accumulation, fp16 out).
//
// Structure (what round 2 measured is in DESIGN_NOTEBOOK.md 4.1; short form):
//   * unit = 128 weight rows x 128 k; stream-K / tile-group split with static tile ownership (tail segment
//     first, head segment last; write-through slabs + counters; waits only point at lower-numbered
//     workgroups);
//   * 12 waves per workgroup, one workgroup per CU.  Waves 8-11 are LOADERS and do nothing but LDS-DMA
//     (global_load_lds: no registers, no ds_write, no permutes): waves 8/9 copy a unit's 8 KB of packed
//     weights + 1 KB of (s, -z*s) pairs into a 6-slot ring five units ahead, waves 10/11 its [64 x 128]
//     activation tile into a 4-slot ring three units ahead (XOR-swizzled through the per-lane SOURCE
//     address so that the consumers' 16-byte fragment reads are bank-conflict free).  The two streams live
//     in different waves on purpose: a wave's memory operations return in order, so L2-resident
//     activations queued behind HBM weight loads in one wave inherit the HBM latency (measured with the
//     self-loading variant of this kernel: compute 0.69 us/unit, memory 0.48, together 1.0 -- a wave blocked
//     on a full memory queue cannot issue its MFMAs);
//   * waves 0-7 are CONSUMERS (4 row groups x 2 k-halves): 10 LDS reads, the exact nibble unpack + fp16
//     affine map (13 VALU per 8 weights) and 8 MFMA 32x32x16 per unit, no global memory instruction at all
//     in the loop.  The weights are stored in the order the MFMA wants them (the packer arranges the
//     nibbles so that the dequantised pairs come out in natural k order): one contiguous KB per wave and
//     unit, nothing is permuted on the way.  All operands of unit u+1 are read into a second register set
//     while unit u is multiplied (bare s_barrier, reads stay in flight across it);
//   * no unit table, no prologue barrier before the first loads: the unit sequence is three scalar
//     segments walked by scalar cursors;
//   * one s_barrier per unit for all twelve waves; the loaders wait (counted vmcnt) for unit u+2 before
//     barrier u, so everything a consumer touches after a barrier has landed;
//   * after the k-half reduction each of the two waves of a row group finishes ONE 32-row half of the
//     batch (flush, merge and epilogue split two ways).
#include <stdlib.h>

#include "common.h"
#include "gemm_debug.h"  // V3_TL / V3_TLC / V3_TL_FENCE: in-kernel stamps of -DV3_TIMELINE builds (empty in the product build)
#include "gemm_w4_common.h"

#define V3_BN 128
#define V3_BM 64
#define V3_CK 128
#define V3_THREADS 768
// LDS rings per tile width (NF = 128-row blocks per tile).  NF = 1: weights + scales 6 slots filled 5 units
// ahead, activations 4 slots / 3 ahead, consumers read one unit ahead of their MFMAs.  NF = 2 (256-row tiles,
// one activation tile feeds two weight blocks): weights 4 slots / 3 ahead, activations 3 slots / 2 ahead, the
// consumers read a unit's operands at its start (a step is twice as long, the LDS latency is paid once).
template <int NF> struct V3Ring;
#ifndef V3_RW1
#define V3_RW1 6
#endif
template <> struct V3Ring<1> { static constexpr int RW = V3_RW1, DW = V3_RW1 - 1, RX = 4, DX = 3, RA = 1; };
template <> struct V3Ring<2> { static constexpr int RW = 4, DW = 3, RX = 3, DX = 2, RA = 0; };
#define V3_MAX_SLOTS 12
#define V3_FRAG 1024                  // floats of one (row group, batch half) partial: 16 per lane
#define V3_SLAB (V3_BN * V3_BM)       // floats per (tile, contributor)
#define V3_W_BLOCK 9216               // 8 KB of packed weights (one KB per consumer wave) + 1 KB of scale pairs
#define V3_X_SLOT 16384               // [64 rows][16 x 16 B], slot j of row r stored at j ^ (r & 15)
template <int NF>
struct V3Lds {
  static constexpr int W_SLOT = NF * V3_W_BLOCK;
  static constexpr int OFF_W = 0;
  static constexpr int OFF_X = V3Ring<NF>::RW * W_SLOT;
  static constexpr int OFF_R = OFF_X + V3Ring<NF>::RX * V3_X_SLOT;  // k-half exchange: 4 row groups x 2 batch halves x 4 KB
  static constexpr int BYTES = OFF_R + 8 * 4096;
  static_assert(BYTES <= 160 * 1024, "LDS");
};
#define V3_SPIN_LIMIT (1 << 24)  // ~10 s of polling, then the launch aborts (never a silent partial sum)

struct V3Params {
  uint16_t* out;
  const uint16_t* x;
  const void* wp;  // packed weights  [tile][chunk][wave 8][lane 64] x 16 B
  const void* sp;  // packed scales   [K/g][N] x 8 B: (s, s), (-z*s, -z*s) fp16 pairs (ll_w4a16_pack_scales)
  const uint16_t* bias;
  float* workspace;
  int32_t* counters;
  int64_t m, n, k;
  int64_t x_stride;
  uint32_t x_cstride;  // bytes between the activation tiles of consecutive 128-k chunks (256 for a row-major [m][k] matrix)
  uint32_t w_bytes, s_bytes, x_bytes;
  int nblocks, chunks, total_units, upw, slots;
  int gt, gbase, grem, glead;  // tile-group split, see v3_plan
  int err_idx;  // index of the launch's error word in `counters` (just past the merge counters)
  int xcd_shift;  // >= 0 (split-K partial mode, gt = 1 << xcd_shift <= 8, grid % 8 == 0): workgroup b = 8 q + x -- on XCD x under
                  // the round-robin dispatch -- takes k-slice x % gt of tile q * (8 / gt) + x / gt, so one XCD's L2 only ever sees
                  // ONE k-slice of the activation matrix (1 / gt of it) instead of all of it; -1: b = tile * gt + slice
  uint32_t chunks_magic, gt_magic, upw_magic;  // ceil(2^32 / d): x / d = umulhi(x, magic) for the unit / workgroup indices of a launch (< 2^20)
  int gshift;                  // log2(group_size / 128)
  int epi;                     // 0: out[m, n];  1: rows are (gate_j, up_j) pairs -> out[m, n/2] = swiglu;  2: fp32 split-K partials [slot][m][n]
  V3_DEBUG_FIELDS
};


// The workgroup's unit sequence: up to three segments (tail of the last tile, the full tiles, head of the
// first tile), each a run of consecutive chunks that wraps into the next tile.  A walker keeps (tile, chunk)
// and two countdowns; the per-unit path is three scalar adds and two compares, the fix-up at the end of a tile
// or segment is a (rare) branch.
struct V3Seq {
  int chunks;
  int t0, c0, n0, t1, c1, n1, t2, c2, n2;
};
struct V3Walk {
  int t, c, seg_left, tile_left, seg;
};
__device__ __forceinline__ V3Walk v3_walk_begin(const V3Seq& q) { return V3Walk{q.t0, q.c0, q.n0, q.chunks - q.c0, 0}; }
// true when the unit the walker stands on ends a tile segment (last chunk of the tile or of the segment)
__device__ __forceinline__ bool v3_walk_ends(const V3Walk& w) { return (w.tile_left == 1) | (w.seg_left == 1); }
__device__ __forceinline__ void v3_walk_next(V3Walk& w, const V3Seq& q) {
  const bool ends = v3_walk_ends(w);
  w.c += 1;
  w.tile_left -= 1;
  w.seg_left -= 1;
  if (ends) {
    if (w.seg_left == 0) {
      w.seg += 1;
      const int s = w.seg;
      w.t = s == 1 ? q.t1 : q.t2;
      w.c = s == 1 ? q.c1 : (s == 2 ? q.c2 : 0);
      const int n = s == 1 ? q.n1 : (s == 2 ? q.n2 : 0);
      w.seg_left = n == 0 ? 0x40000000 : n;  // past the end: stays there
    } else {
      w.t += 1;
      w.c = 0;
    }
    w.tile_left = q.chunks - w.c;
  }
}

// One statement per unit and loader wave (M0 saved / restored once, the pieces' LDS addresses in scalar registers): a weight
// loader's four 1-KB pieces + two 256-B scale pieces, an activation loader's eight 1-KB pieces.
#define V3_STR2(X) #X
#define V3_STR(X) V3_STR2(X)
#if (V3_NT_W & 1)
#define V3_W_POLICY " nt"
#else
#define V3_W_POLICY ""
#endif
#if (V3_NT_W & 2)
#define V3_S_POLICY " nt"
#else
#define V3_S_POLICY ""
#endif
__device__ __forceinline__ void v3_dma_w(uint32_t dw, uint32_t ds, const void* wb, const void* sb, const uint32_t (&voff)[8]) {
  uint32_t keep;
  dw = __builtin_amdgcn_readfirstlane(dw);
  ds = __builtin_amdgcn_readfirstlane(ds);
  wb = v3_uniform_ptr(wb);
  sb = v3_uniform_ptr(sb);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %13" V3_W_POLICY "\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %13" V3_W_POLICY "\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %13" V3_W_POLICY "\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %13" V3_W_POLICY "\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %11, %14" V3_S_POLICY "\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %12, %14" V3_S_POLICY "\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(dw), "s"(dw + 1024), "s"(dw + 2048), "s"(dw + 3072), "s"(ds), "s"(ds + 256), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]),
        "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "s"(wb), "s"(sb)
      : "memory");
}
__device__ __forceinline__ void v3_dma_x(uint32_t dx, const void* xb, const uint32_t (&voff)[8]) {
  uint32_t keep;
  dx = __builtin_amdgcn_readfirstlane(dx);
  xb = v3_uniform_ptr(xb);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %17\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %17\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %11, %17\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %12, %17\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %13, %17\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %14, %17\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %15, %17\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %16, %17\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(dx), "s"(dx + 1024), "s"(dx + 2048), "s"(dx + 3072), "s"(dx + 4096), "s"(dx + 5120), "s"(dx + 6144), "s"(dx + 7168),
        "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7]), "s"(xb)
      : "memory");
}
// Units a loader may request between the prologue barrier P0 and the barrier that ends unit 0 (the consumers wait there for
// the loaders' ISSUE of these requests, ~0.65 us per unit and loader wave): A/B knob, see DESIGN_NOTEBOOK.md 4.3
#ifndef V3_START_FILL
#define V3_START_FILL 8
#endif
// at most k units of OPS memory operations each may still be in flight
template <int OPS>
__device__ __forceinline__ void v3_wait_units(int k) {
  if (k <= 0) v3_vmcnt<0>();
  else if (k == 1) v3_vmcnt<OPS>();
  else if (k == 2) v3_vmcnt<2 * OPS>();
  else if (k == 3 || 4 * OPS > 63) v3_vmcnt<3 * OPS>();
  else v3_vmcnt<(4 * OPS > 63 ? 3 : 4) * OPS>();
}
__device__ __forceinline__ void v3_barrier() { asm volatile("s_barrier" ::: "memory"); }

// One loader wave.  KIND 0: weight pieces 4L..4L+3 (1 KB each) + scale quarters 2L, 2L+1 (256 B each) of every
// unit, V3_DW units ahead into a V3_RW-slot ring.  KIND 1: activation pieces 8L..8L+7 (4 rows x 256 B each),
// V3_DX units ahead into a V3_RX-slot ring; LDS image row r, 16-B slot j <- source slot j ^ (r & 15).
template <int KIND, int NF>
__device__ __forceinline__ void v3_loader(const V3Params& p, const V3Seq& q, int cnt, int lane, int L) {
  using RG = V3Ring<NF>;
  using LD = V3Lds<NF>;
  constexpr int OPS = KIND == 0 ? 6 * NF : 8;
  constexpr int D = KIND == 0 ? RG::DW : RG::DX;
  constexpr int R = KIND == 0 ? RG::RW : RG::RX;
  constexpr int AHEAD = 1 + RG::RA;  // after barrier u the consumers may touch units <= u + AHEAD
  static_assert(D - AHEAD <= 4 && D - AHEAD >= 1 && D <= R - 1 && (D - AHEAD) * OPS <= 63, "ring depth");
  uint32_t voff[8];
  if constexpr (KIND == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) voff[j] = (uint32_t)(lane * 16 + (4 * L + j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) voff[4 + j] = (uint32_t)(lane * 4 + (2 * L + j) * 256);
    voff[6] = voff[7] = 0;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t r = (8 * L + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (int)(r & 15);
      if (r >= p.m) r = p.m - 1;  // rows >= M feed only unstored outputs
      voff[j] = (uint32_t)(r * p.x_stride * 2 + slot * 16);
    }
  }
  V3Walk lc = v3_walk_begin(q);
  int issued = 0;
  uint32_t dst = KIND == 0 ? LD::OFF_W : LD::OFF_X;  // ring slot the next unit goes to
  auto issue = [&]() {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {  // the tile's 128-row blocks lc.t * NF + f
        const int blk = lc.t * NF + f;
        const char* wb = (const char*)p.wp + (size_t)(uint32_t)((blk * q.chunks + lc.c) * (V3_BN * V3_CK / 2));
        const char* sb = (const char*)p.sp + (size_t)(uint32_t)(((lc.c >> p.gshift) * (int)p.n + blk * V3_BN) * 8);
        v3_dma_w(dst + f * V3_W_BLOCK + 4 * L * 1024, dst + f * V3_W_BLOCK + 8192 + 2 * L * 256, wb, sb, voff);
      }
    } else {
      const char* xb = (const char*)p.x + (size_t)((uint32_t)lc.c * p.x_cstride);
      v3_dma_x(dst + 8 * L * 1024, xb, voff);
    }
    v3_walk_next(lc, q);
    ++issued;
    if constexpr (KIND == 0) dst = dst + LD::W_SLOT == LD::OFF_W + R * LD::W_SLOT ? LD::OFF_W : dst + LD::W_SLOT;
    else dst = dst + V3_X_SLOT == LD::OFF_X + R * V3_X_SLOT ? LD::OFF_X : dst + V3_X_SLOT;
  };
  const int wv = KIND == 0 ? 8 + L : 10 + L;  // (timeline builds)
  (void)wv;
  // Prologue (round 3): the whole ring depth is requested at once, in unit order -- a wave's operations return in order, so
  // unit 0 lands first -- and the consumers start on unit 0 as soon as IT has landed (barrier P0); with operands read one
  // unit ahead (RA) they multiply unit 0 while units 1, 2 land and meet the loaders again at B0.  (Round 2 requested two
  // units, waited for both, and refilled the ring two units per step: first unit finished 4.2-4.8 us after entry.)
  const int pre = cnt < 1 + AHEAD ? cnt : 1 + AHEAD;  // what the consumers' first two steps need; the rest of the ring after P0
  for (int i = 0; i < pre; ++i) issue();
  V3_TL(2)
  v3_wait_units<OPS>(issued - 1);  // unit 0 has landed
  v3_barrier();                    // P0
  V3_TL(3)
  V3Walk cc = v3_walk_begin(q);
  int u0 = 0;
  if constexpr (RG::RA) {
    // the consumers' first step: unit 0 alone (no operands read ahead)
    int fill = cnt < D ? cnt : D;
    if (fill > issued + V3_START_FILL) fill = issued + V3_START_FILL;
    while (issued < fill) issue();
    const int need = cnt < 1 + AHEAD ? cnt : 1 + AHEAD;
    v3_wait_units<OPS>(issued - need);
    v3_barrier();  // B0
    if (v3_walk_ends(cc)) v3_barrier();  // k-half exchange (128-row tiles: one barrier)
    v3_walk_next(cc, q);
    u0 = 1;
  }
  for (int u = u0; u < cnt; ++u) {
    const int want = cnt < u + 1 + D ? cnt : u + 1 + D;
    if (issued < want) issue();
    if (issued < want && (u > 0 || V3_START_FILL >= 2)) issue();
    if (u < 12) { V3_TL(4 + 3 * u) }
    const int need = cnt < u + 1 + AHEAD ? cnt : u + 1 + AHEAD;
    v3_wait_units<OPS>(issued - need);
    if (u < 12) { V3_TL(5 + 3 * u) }
    v3_barrier();
    if (u < 12) { V3_TL(6 + 3 * u) }
    if (v3_walk_ends(cc)) {  // the consumers' k-half exchange: 1 barrier, or 3 for 256-row tiles with two batch halves
      const int nb = NF == 1 ? 1 : (p.m > 32 ? 3 : 1);
      for (int f = 0; f < nb; ++f) v3_barrier();
    }
    v3_walk_next(cc, q);
  }
}

struct V3One { static constexpr int value = 1; };
struct V3Two { static constexpr int value = 2; };

template <int MT, int NF>
struct V3Ops {  // everything a consumer wave needs for one unit
  u32x4 w[NF];
  u32x2 s[NF];
  f16x8 a[4][MT];
};

__device__ __forceinline__ int v3_div(int x, uint32_t magic, int d) {
  (void)d;
  return magic ? (int)__umulhi((uint32_t)x, magic) : x;  // magic 0: d = 1
}

template <int MT, int NF>
__global__ __launch_bounds__(V3_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgemm3_kernel(const V3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = p.chunks;
  V3_TL(0)
  V3_TLC(61)
  int ub, ue;
  int my_slice = -1;
  if (p.gt) {
    int gtile, j;
    if (p.xcd_shift >= 0) {
      const int x = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
      j = x & (p.gt - 1);
      gtile = q * (8 >> p.xcd_shift) + (x >> p.xcd_shift);
      my_slice = j;
    } else {
      gtile = v3_div((int)blockIdx.x, p.gt_magic, p.gt);
      j = (int)blockIdx.x - gtile * p.gt;
    }
    const int lo = j * p.gbase + (j < p.grem ? j : p.grem);
    ub = gtile * chunks + lo;
    ue = j == p.gt - 1 ? (gtile + 1) * chunks : ub + p.gbase + (j < p.grem ? 1 : 0);
  } else {
    ub = blockIdx.x * p.upw;
    ue = ub + p.upw;
    if (ue > p.total_units) ue = p.total_units;
  }
  if (ub >= ue) return;
  const int cnt = ue - ub;
  const int tA = v3_div(ub, p.chunks_magic, chunks), cA = ub - tA * chunks;
  const int tZ = v3_div(ue - 1, p.chunks_magic, chunks), cZ = (ue - 1) - tZ * chunks;
  int LT = 0, LH = cnt;  // a range inside one tile runs as a single "head" segment
  if (tA != tZ) {
    LT = (cZ != chunks - 1) ? cZ + 1 : 0;
    LH = (cA != 0) ? chunks - cA : 0;
  }
  const int NFU = cnt - LT - LH;  // units of whole tiles
  const int tF = tA + ((LH > 0 && tA != tZ) ? 1 : 0);
  // segments in execution order, empty ones squeezed out (selects only: a runtime-indexed array would live in scratch)
  const bool hasT = LT > 0, hasF = NFU > 0;
  V3Seq q;
  q.chunks = chunks;
  q.t0 = hasT ? tZ : (hasF ? tF : tA); q.c0 = hasT ? 0 : (hasF ? 0 : cA); q.n0 = hasT ? LT : (hasF ? NFU : LH);
  q.t1 = (hasT && hasF) ? tF : tA; q.c1 = (hasT && hasF) ? 0 : cA; q.n1 = hasT ? (hasF ? NFU : LH) : (hasF ? LH : 0);
  q.t2 = tA; q.c2 = cA; q.n2 = (hasT && hasF) ? LH : 0;
  V3_TL(1)

  if (wv >= 8) {
    // ======================================== loaders ======================================== //
    if (wv < 10) v3_loader<0, NF>(p, q, cnt, lane, wv - 8);
    else v3_loader<1, NF>(p, q, cnt, lane, wv - 10);
    return;
  }

  // ======================================= consumers ======================================= //
  using RG = V3Ring<NF>;
  using LD = V3Lds<NF>;
  const int ng = wv & 3, kh = wv >> 2;
  const int nl = lane & 31, h = lane >> 5;
  const int w_off = wv * 1024 + lane * 16;
  const int s_off = 8192 + (ng * 32 + nl) * 8;
  int x_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x_off[j] = nl * 256 + (((kh * 8 + h * 4 + j) ^ (nl & 15)) * 16);

  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  // (128-row block f, batch half mt) -> acc{2f + mt}.  Separate values, not an array: a wave-uniform choice
  // between them must stay a select (a runtime-indexed array would live in scratch)
  f32x16 acc0, acc1, acc2, acc3;
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  };
  zero_acc();

  // contribution counters of a parked partial (pend_n consecutive ones: one per batch half this wave finished),
  // posted once its write-through stores have landed (the consumers have no other memory operation in flight:
  // vmcnt(0) waits for exactly those stores)
  int32_t* pend_ctr = nullptr;
  int pend_val = 0, pend_n = 0;
  int pending = 0;
  auto post_pending = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < pend_n) __hip_atomic_fetch_add(pend_ctr + lane, pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = 0;
  };

  // Finish NM (1 or 2) 32-row batch halves mt0, mt0 + 1 of row group ng, 128-row block f of tile t, whose chunks
  // [c_lo, c_hi] this workgroup has just summed into v0 (and v1) -- k-halves already added.
  auto flush = [&](f32x16& v0, f32x16& v1, auto nm_tag, int mt0, int t, int f, int c_lo, int c_hi) {
    constexpr int NM = decltype(nm_tag)::value;
    const int w0 = p.gt ? t * p.gt : v3_div(t * chunks, p.upw_magic, p.upw);  // first contributor of the tile
    const int slot = my_slice >= 0 ? my_slice : (int)blockIdx.x - w0;
    const int blk = t * NF + f;  // 128-row block
    auto vsel = [&](int i) -> f32x16& { return i == 0 ? v0 : v1; };
    if (p.epi == 2) {
      // split-K partial mode: every workgroup of the tile group leaves its fp32 partial [slot][m][n] for the
      // consumer kernel (ll_skip_rmsnorm_partials) to add up -- no counters, no polling, no owner
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const f32x16& v = vsel(i);
        const int64_t mrow = nl + (mt0 + i) * 32;
        if (mrow < p.m) {
          float* dst = reinterpret_cast<float*>(p.out) + ((int64_t)slot * p.m + mrow) * p.n + (int64_t)blk * V3_BN + ng * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *reinterpret_cast<f32x4*>(dst + 8 * g) = o;  // plain stores: write-through (sc1) and nt both measured slower (rounds 3, 4)
          }
        }
      }
      return;
    }
    bool poison = false;
    int32_t* ctr = &p.counters[(blk * 4 + ng) * 2 + mt0];
    auto slab = [&](int sq, int mt) { return p.workspace + ((((int64_t)blk * p.slots + sq) * 4 + ng) * 2 + mt) * V3_FRAG; };
    if (c_hi != chunks - 1) {
      // contributor: park the partials in this workgroup's slab (counters follow, see post_pending)
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const f32x16& v = vsel(i);
        float* ws = slab(slot, mt0 + i);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          float* dst = ws + (g * 64 + lane) * 4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(o) : "memory");
        }
      }
      V3_TL(58)
      pend_ctr = ctr;
      pend_val = c_hi - c_lo + 1;
      pend_n = NM;
      pending = 1;
      return;
    }
    if (c_lo != 0) {
      // owner: chunks [0, c_lo) were summed by the `slot` lower-numbered contributors
      bool arrived = false;
      for (int spin = 0; spin < V3_SPIN_LIMIT; ++spin) {
        int seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (NM == 2) {
          const int s1 = __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          seen = seen < s1 ? seen : s1;
        }
        if (__builtin_amdgcn_readfirstlane(seen) >= c_lo) {
          arrived = true;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      if (arrived) {
        if (lane < NM) __hip_atomic_store(ctr + lane, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // A contributor that has not delivered after ~10 s of polling is not coming (its workgroup was never resident, or
        // died).  Never a silent partial sum: the tile is written as NaN -- the poison reaches the logits --
        // and the merge counters are LEFT non-zero, plus the launch's error word (counters[err_idx]): the counter buffer is
        // all zero at rest by construction, so "any non-zero word after a synchronisation" is the sticky error the host
        // checks (DecodeEngine._check_device_errors, _lib.gemm_scratch_error).  (A __builtin_trap() here cost the kernel two
        // user SGPRs for the queue pointer and 15-25 % of its speed through spills in the unit loop: measured, removed.)
        if (lane == 0) __hip_atomic_store(p.counters + p.err_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poison = true;
      }
      V3_TL(52)
      // slabs were written through (sc1) before their counter: coherent (sc1) loads, no acquire fence.
      // Sixteen loads in flight per round trip (4 slabs x 1 batch half or 2 x 2); the tail of the last round is
      // masked to +0.
      constexpr int SLB = 4 / NM;
      for (int sl = 0; sl < slot; sl += SLB) {
        i32x4 va[SLB][NM][4];
#pragma unroll
        for (int j = 0; j < SLB; ++j) {
          const int sq = sl + j < slot ? sl + j : sl;
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            const float* src = slab(sq, mt0 + i) + lane * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(va[j][i][g]) : "v"(src + g * 256) : "memory");
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < SLB; ++j)
#pragma unroll
          for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(va[j][i][g]));  // uses stay below the wait
#pragma unroll
        for (int j = 0; j < SLB; ++j) {
          const int keep = sl + j < slot ? -1 : 0;
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            f32x16& v = vsel(i);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * g + e] += __int_as_float(va[j][i][g][e] & keep);
          }
        }
      }
    }
    V3_TL(53)
    const bool has_bias = p.bias != nullptr;
    // Output stores (round 3): a lane holds, for its batch row, the rows 8g + 4h .. + 3 of the wave's 32 (g = 0..3) -- four
    // separate 8-byte pieces (4 bytes after swiglu).  Lanes nl and nl + 32 (h = 0 / 1) hold the interleaving pieces of the
    // SAME batch row: one v_permlane32_swap per register hands lane h = 0 both halves of groups 0, 1 and lane h = 1 both
    // halves of groups 2, 3, so every lane writes 16 contiguous bytes per store (4x fewer, 2-4x wider store instructions:
    // the tail of the fused gate|up launch was store-issue bound).
    auto swap32 = [](uint32_t& a, uint32_t& b) {  // lanes h = 1 of `a` <-> lanes h = 0 of `b`
      const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
      a = r[0];
      b = r[1];
    };
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const f32x16& v = vsel(i);
      const int64_t mrow = nl + (mt0 + i) * 32;
      const bool row_ok = mrow < p.m;
      uint32_t lo[4], hi[4];  // per g: fp16 pairs (o0, o1), (o2, o3)
      uint32_t sw[4];         // per g: swiglu pair
      // (round 4: straight-line code -- the four bias halves of a group as one 8-byte load behind ONE branch, the NaN poison as
      // a select; the per-element branches and the libm sigmoid of round 3 made this epilogue 2.6 us of dependent VALU chains)
      float fv[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) fv[g][e] = v[4 * g + e];
      if (has_bias) {
        uint2 bb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bb[g] = *reinterpret_cast<const uint2*>(p.bias + (int64_t)blk * V3_BN + ng * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          fv[g][0] += f16_bits_to_f32((uint16_t)(bb[g].x & 0xffffu));
          fv[g][1] += f16_bits_to_f32((uint16_t)(bb[g].x >> 16));
          fv[g][2] += f16_bits_to_f32((uint16_t)(bb[g].y & 0xffffu));
          fv[g][3] += f16_bits_to_f32((uint16_t)(bb[g].y >> 16));
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = poison ? (uint16_t)0x7e00 : f32_to_f16_bits(fv[g][e]);
        lo[g] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        hi[g] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        if (p.epi) {
          // weight rows 2j / 2j+1 are gate_j / up_j: both land in this lane.  Same arithmetic as the
          // stand-alone kernels: the two GEMM outputs rounded to fp16, then silu(g) * u in fp32.
          const float g0 = f16_bits_to_f32(o[0]), u0 = f16_bits_to_f32(o[1]);
          const float g1 = f16_bits_to_f32(o[2]), u1 = f16_bits_to_f32(o[3]);
          const uint32_t s0 = f32_to_f16_bits(ll_silu_mul_f32(g0, u0));
          const uint32_t s1 = f32_to_f16_bits(ll_silu_mul_f32(g1, u1));
          sw[g] = s0 | (s1 << 16);
        }
      }
      V3_TL(58)
      const int64_t n0 = (int64_t)blk * V3_BN + ng * 32 + 16 * h;  // first of the 16 rows this lane stores after the swap
      if (p.epi) {
        swap32(sw[0], sw[2]);
        swap32(sw[1], sw[3]);
        if (row_ok) *reinterpret_cast<u32x4*>(p.out + mrow * (p.n >> 1) + (n0 >> 1)) = u32x4{sw[0], sw[2], sw[1], sw[3]};
      } else {
        swap32(lo[0], lo[2]);
        swap32(hi[0], hi[2]);
        swap32(lo[1], lo[3]);
        swap32(hi[1], hi[3]);
        if (row_ok) {
          *reinterpret_cast<u32x4*>(p.out + mrow * p.n + n0) = u32x4{lo[0], hi[0], lo[2], hi[2]};
          *reinterpret_cast<u32x4*>(p.out + mrow * p.n + n0 + 8) = u32x4{lo[1], hi[1], lo[3], hi[3]};
        }
      }
      V3_TL(59)
    }
  };

  // End of a tile segment: the two k-halves of a row group exchange partial sums through LDS (4 KB per wave).
  //   128-row tiles: wave kh ends up with the complete sums of batch half kh (MT = 2) or kh = 0 with everything (MT = 1);
  //   256-row tiles: wave kh ends up with BOTH batch halves of 128-row block kh, so the two blocks' flushes (slab
  //   stores, or counter wait + slab loads + epilogue) run side by side in the two waves instead of one after the other.
  auto segment_end = [&](int t, int c_lo, int c_hi) {
    float* red = reinterpret_cast<float*>(lds + LD::OFF_R) + ng * (2 * V3_FRAG);
    V3_TL(50)
    auto put = [&](const f32x16& a, int half) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(red + half * V3_FRAG + (g * 64 + lane) * 4) =
            f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    };
    auto get_add = [&](f32x16& a, int half) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(red + half * V3_FRAG + (g * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * g + e] += o[e];
      }
    };
    auto xbarrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if constexpr (NF == 1) {
      if constexpr (MT == 2) {
        f32x16 give, v;  // one copy of the exchange / flush code for both waves of the row group
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          give[r] = kh == 0 ? acc1[r] : acc0[r];
          v[r] = kh == 0 ? acc0[r] : acc1[r];
        }
        put(give, 1 - kh);
        xbarrier();
        V3_TL(51)
        get_add(v, kh);
        flush(v, v, V3One{}, kh, t, 0, c_lo, c_hi);
      } else {
        if (kh == 1) put(acc0, 0);
        xbarrier();
        if (kh == 0) {
          get_add(acc0, 0);
          flush(acc0, acc0, V3One{}, 0, t, 0, c_lo, c_hi);
        }
      }
    } else {
      // wave kh keeps block kh: round A moves the batch-half-0 partials, round B the batch-half-1 ones
      f32x16 give, v0, v1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        give[r] = kh == 0 ? acc2[r] : acc0[r];
        v0[r] = kh == 0 ? acc0[r] : acc2[r];
      }
      put(give, kh);
      xbarrier();
      V3_TL(51)
      get_add(v0, 1 - kh);
      if constexpr (MT == 2) {
        xbarrier();  // the partner has read round A before round B overwrites the buffer
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          give[r] = kh == 0 ? acc3[r] : acc1[r];
          v1[r] = kh == 0 ? acc1[r] : acc3[r];
        }
        put(give, kh);
        xbarrier();
        get_add(v1, 1 - kh);
        flush(v0, v1, V3Two{}, 0, t, kh, c_lo, c_hi);
      } else {
        flush(v0, v0, V3One{}, 0, t, kh, c_lo, c_hi);
      }
    }
    zero_acc();
  };

  int done = 0;
  auto read_ops = [&](V3Ops<MT, NF>& o, int wbase, int xbase) {  // ring slot byte offsets
    const unsigned char* wb = lds + wbase;
    const unsigned char* xb = lds + xbase;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      o.w[f] = *reinterpret_cast<const u32x4*>(wb + f * V3_W_BLOCK + w_off);
      o.s[f] = *reinterpret_cast<const u32x2*>(wb + f * V3_W_BLOCK + s_off);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) o.a[j][mt] = *reinterpret_cast<const f16x8*>(xb + x_off[j] + mt * 32 * 256);
  };
  auto compute = [&](const V3Ops<MT, NF>& o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const uint32_t word = j == 0 ? o.w[f].x : j == 1 ? o.w[f].y : j == 2 ? o.w[f].z : o.w[f].w;
        const f16x8 wfrag = v3_dequant(word, o.s[f].x, o.s[f].y, magic);
        if (f == 0) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, o.a[j][0], acc0, 0, 0, 0);
          if constexpr (MT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, o.a[j][1], acc1, 0, 0, 0);
        } else {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, o.a[j][0], acc2, 0, 0, 0);
          if constexpr (MT == 2) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, o.a[j][1], acc3, 0, 0, 0);
        }
      }
    }
  };

  V3Walk cc = v3_walk_begin(q);
  int seg_lo = cc.c;
  int wnext = LD::OFF_W + RG::RA * LD::W_SLOT, xnext = LD::OFF_X + RG::RA * V3_X_SLOT;  // ring slots of the next unit to READ
  V3Ops<MT, NF> opA, opB;
  v3_barrier();  // P0: unit 0 has landed
  V3_TL(3)
  if constexpr (RG::RA) {
    // first step: unit 0 is multiplied as soon as it is there (its operands are read now, not one unit ahead); the loaders
    // meet the consumers again at B0 with units <= 2 landed, and the steady loop starts at unit 1
    read_ops(opA, LD::OFF_W, LD::OFF_X);
    compute(opA);
    v3_barrier();  // B0
    V3_TL(4)
    const bool se0 = v3_walk_ends(cc);
    if (se0) segment_end(cc.t, seg_lo, cc.c);
    v3_walk_next(cc, q);
    if (se0) seg_lo = cc.c;
    done = 1;
    if (done < cnt) {
      read_ops(opA, wnext, xnext);
      wnext += LD::W_SLOT;
      xnext += V3_X_SLOT;
    }
  }

  // One unit.  RA = 1 (128-row tiles): the operands of unit `done` are in CUR (read during the previous unit); the
  // operands of the next unit are read into NXT first -- they landed before the barrier that ended the previous
  // unit -- and stay in flight across this unit's barrier.  RA = 0 (256-row tiles): a unit's operands are read at
  // its start.
#define V3_STEP(CUR, NXT)                                                                   \
  {                                                                                         \
    if (done == 5) { V3_TLC(44) }                                                           \
    read_ops(RG::RA ? NXT : CUR, wnext, xnext);                                             \
    __builtin_amdgcn_sched_barrier(0); /* keep the reads up here (hipcc sinks them to their use otherwise) */ \
    if (done == 5) { V3_TLC(45) }                                                           \
    wnext = wnext + LD::W_SLOT == LD::OFF_W + RG::RW * LD::W_SLOT ? LD::OFF_W : wnext + LD::W_SLOT; \
    xnext = xnext + V3_X_SLOT == LD::OFF_X + RG::RX * V3_X_SLOT ? LD::OFF_X : xnext + V3_X_SLOT; \
    compute(CUR);                                                                           \
    if constexpr (!RG::RA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the slot is free once the barrier is passed */ \
    V3_TL_FENCE(CUR)                                                                        \
    if (done == 5) { V3_TLC(46) }                                                           \
    v3_barrier();                                                                           \
    if (done == 5) { V3_TLC(47) }                                                           \
    if (done < 40) { V3_TL(4 + done) }                                                      \
    if (pending) post_pending(); /* the previous segment's slab stores are a unit old */    \
    const bool se_ = v3_walk_ends(cc);                                                      \
    if (se_) segment_end(cc.t, seg_lo, cc.c);                                               \
    v3_walk_next(cc, q);                                                  \
    if (se_) seg_lo = cc.c;                                                                 \
    if (done == 5) { V3_TLC(48) }                                                           \
    ++done;                                                                                 \
  }
  if (done < cnt) for (;;) {
    V3_STEP(opA, opB)
    if (done >= cnt) break;
    if constexpr (RG::RA) {
      V3_STEP(opB, opA)
      if (done >= cnt) break;
    }
  }
#undef V3_STEP
  if (pending) post_pending();
  V3_TL(60)
  V3_TLC(62)
}

// ---------------------------------------------------------------------------------- //
// load-time packers
// ---------------------------------------------------------------------------------- //
// Bit-exact permutation of the reference layout (qweight [N, K/8] int32, nibble j of word i = k 8i + j,
// w4a16.py:99-105) into the stream order of wgemm3_kernel: block (tile = n / 128, chunk = k / 128) is 8 KB
// = [wave 8][lane 64] x 16 B; wave (ng = w & 3, kh = w >> 2), lane (nl = l & 31, h = l >> 5) owns the four
// source words c*16 + kh*8 + h*4 + 0..3 of row tile*128 + ng*32 + nl; inside each word the even nibbles go to
// positions 0..3 and the odd ones to 4..7.
__device__ __forceinline__ uint32_t v3_nib_perm(uint32_t w) {
  uint32_t ev = w & 0x0F0F0F0Fu, od = (w >> 4) & 0x0F0F0F0Fu;
  ev = (ev | (ev >> 4)) & 0x00FF00FFu;
  ev = (ev | (ev >> 8)) & 0x0000FFFFu;
  od = (od | (od >> 4)) & 0x00FF00FFu;
  od = (od | (od >> 8)) & 0x0000FFFFu;
  return ev | (od << 16);
}

// (s, -z*s) as fp16 pairs, transposed to [group][row]: a unit's 128 rows become 1 KB contiguous (the fp32 [N, K/g] grids would
// cost 256 scattered 4-byte requests per unit -- twice the weight stream's).  fp32 product, one rounding each.
__global__ void w4a16_pack_scales_kernel(uint2* packed, const float* scales, const float* zeros, int64_t n,
                                         int64_t groups, int64_t s_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * groups) return;
  const int64_t g = i / n, r = i - g * n;
  const float s = scales[r * s_stride + g], z = zeros[r * s_stride + g];
  const uint32_t hs = f32_to_f16_bits(s), hz = f32_to_f16_bits(-z * s);
  uint2 o;
  o.x = hs | (hs << 16);
  o.y = hz | (hz << 16);
  packed[i] = o;
}

extern "C" int ll_w4a16_pack_scales(void* packed, const float* scales, const float* zeros, int64_t n,
                                    int64_t groups, int64_t s_stride_n, void* stream) {
  if (n <= 0 || groups <= 0) return LL_ERR_SHAPE;
  if (!packed || !scales || !zeros) return LL_ERR_ARG;
  const int64_t total = n * groups;
  w4a16_pack_scales_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (uint2*)packed, scales, zeros, n, groups, s_stride_n);
  return LL_LAUNCH_CHECK();
}

__global__ void w4a16_pack_weights_kernel(u32x4* dst, const uint32_t* qw, int64_t total, int chunks, int64_t qw_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63), w = (int)((i >> 6) & 7);
  const int64_t blk = i >> 9;
  const int64_t t = blk / chunks;
  const int c = (int)(blk - t * chunks);
  const int64_t row = t * V3_BN + (w & 3) * 32 + (lane & 31);
  const int word0 = c * 16 + (w >> 2) * 8 + (lane >> 5) * 4;
  const u32x4 s = *reinterpret_cast<const u32x4*>(qw + row * qw_stride + word0);
  u32x4 o;
  o.x = v3_nib_perm(s.x);
  o.y = v3_nib_perm(s.y);
  o.z = v3_nib_perm(s.z);
  o.w = v3_nib_perm(s.w);
  dst[i] = o;
}

extern "C" int ll_w4a16_pack_weights(void* packed, const int32_t* qweight, int64_t n, int64_t k, int64_t qw_stride_n,
                                     void* stream) {
  if (n <= 0 || k <= 0 || n % V3_BN != 0 || k % V3_CK != 0 || qw_stride_n % 4 != 0) return LL_ERR_SHAPE;
  if (!packed || !qweight || !ll_aligned16(packed) || !ll_aligned16(qweight)) return LL_ERR_ARG;
  const int64_t total = n * k / 32;  // 16-byte pieces
  w4a16_pack_weights_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (u32x4*)packed, (const uint32_t*)qweight, total, (int)(k / V3_CK), qw_stride_n);
  return LL_LAUNCH_CHECK();
}

// The inverse permutation (bit-exact): packed -> qweight [N, K/8] in the reference layout.  With it the load-time layout can be
// the ONLY resident copy of the int4 weights (lite_llama_amd/quantization/methods.py::compact): the reference-format tensor
// is rebuilt on demand for the calls the decode engine does not serve (more than 64 rows) and for checkpoint export.
__device__ __forceinline__ uint32_t v3_nib_unperm(uint32_t p) {
  uint32_t ev = p & 0xFFFFu, od = p >> 16;
  ev = (ev | (ev << 8)) & 0x00FF00FFu;
  ev = (ev | (ev << 4)) & 0x0F0F0F0Fu;
  od = (od | (od << 8)) & 0x00FF00FFu;
  od = (od | (od << 4)) & 0x0F0F0F0Fu;
  return ev | (od << 4);
}

__global__ void w4a16_unpack_weights_kernel(uint32_t* qw, const u32x4* src, int64_t total, int chunks, int64_t qw_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63), w = (int)((i >> 6) & 7);
  const int64_t blk = i >> 9;
  const int64_t t = blk / chunks;
  const int c = (int)(blk - t * chunks);
  const int64_t row = t * V3_BN + (w & 3) * 32 + (lane & 31);
  const int word0 = c * 16 + (w >> 2) * 8 + (lane >> 5) * 4;
  const u32x4 s = src[i];
  u32x4 o;
  o.x = v3_nib_unperm(s.x);
  o.y = v3_nib_unperm(s.y);
  o.z = v3_nib_unperm(s.z);
  o.w = v3_nib_unperm(s.w);
  *reinterpret_cast<u32x4*>(qw + row * qw_stride + word0) = o;
}

extern "C" int ll_w4a16_unpack_weights(int32_t* qweight, const void* packed, int64_t n, int64_t k, int64_t qw_stride_n,
                                       void* stream) {
  if (n <= 0 || k <= 0 || n % V3_BN != 0 || k % V3_CK != 0 || qw_stride_n % 4 != 0) return LL_ERR_SHAPE;
  if (!packed || !qweight || !ll_aligned16(packed) || !ll_aligned16(qweight)) return LL_ERR_ARG;
  const int64_t total = n * k / 32;  // 16-byte pieces
  w4a16_unpack_weights_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (uint32_t*)qweight, (const u32x4*)packed, total, (int)(k / V3_CK), qw_stride_n);
  return LL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct V3Plan {
  int nf;  // 128-row blocks per tile
  int nblocks, chunks, total_units, upw, grid, slots;
  int gt, gbase, grem, glead;
  int xcd_shift = -1;  // V3Params::xcd_shift
};

struct V3Knobs {
  int wgs = 0, lead = 4, gt_cap_div = 5, gt_cap = -1, nf = 0;
  int xcd = 8;  // split-K partial launches take a power-of-two split <= this and the XCD-aware map (LL_GEMM3_XCD=0: off)
  int fill = 85;  // ... when the launch still fills this percentage of the CUs (LL_GEMM3_FILL)
  int short_k_xcd = 8;   // LL_GEMM3_SHORTK_XCD: the same cap for projections with a SHORT contraction (<= 32 chunks) and <= 32 tiles
                         // (the attention output projection): fewer k-slices = fewer fp32 planes for the add-and-normalise that
                         // follows, at a GEMM that fills fewer CUs -- A/B knob of the round-4 review's "4-plane o" (DESIGN_NOTEBOOK.md 4.5)
  V3Knobs() {
    if (const char* e = getenv("LL_GEMM3_XCD")) xcd = atoi(e);
    if (const char* e = getenv("LL_GEMM3_FILL")) fill = atoi(e);
    if (const char* e = getenv("LL_GEMM3_SHORTK_XCD")) short_k_xcd = atoi(e);
    if (const char* e = getenv("LL_GEMM3_WGS")) wgs = atoi(e);
    if (const char* e = getenv("LL_GEMM3_LEAD")) lead = atoi(e);
    if (const char* e = getenv("LL_GEMM3_GT")) gt_cap = atoi(e);
    if (const char* e = getenv("LL_GEMM3_GTDIV")) gt_cap_div = atoi(e) > 0 ? atoi(e) : 5;
    if (const char* e = getenv("LL_GEMM3_NF")) nf = atoi(e);
  }
};
static const V3Knobs& v3_knobs() {
  static const V3Knobs k;  // the environment is read once
  return k;
}

static int v3_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

static V3Plan v3_plan(int64_t n, int64_t k, int nf_force = 0, bool partials = false) {
  const V3Knobs& kn = v3_knobs();
  V3Plan pl;
  const int target = kn.wgs > 0 ? kn.wgs : v3_num_cus();  // one persistent 12-wave workgroup per CU
  // 256-row tiles (one activation tile feeds two weight blocks: half the activation traffic and half the
  // per-unit overhead per weight byte) when there are enough of them to keep the split coarse; 128 otherwise
  pl.nf = (n % (2 * V3_BN) == 0 && n / (2 * V3_BN) >= target / 2) ? 2 : 1;
  if (kn.nf == 1 || (kn.nf == 2 && n % (2 * V3_BN) == 0)) pl.nf = kn.nf;
  if (nf_force == 1 || (nf_force == 2 && n % (2 * V3_BN) == 0)) pl.nf = nf_force;
  if (partials) pl.nf = 1;
  pl.nblocks = (int)(n / (V3_BN * pl.nf));
  pl.chunks = (int)(k / V3_CK);
  pl.total_units = pl.nblocks * pl.chunks;
  pl.gt = pl.gbase = pl.grem = pl.glead = 0;
  // Few tiles: every tile is shared by gt workgroups; the last one (the owner) runs `lead` chunks longer
  // than the contributors, so their slabs and counters have landed by the time it merges.
  const int lead = partials ? 0 : kn.lead;  // partial mode has no owner: equal shares
  int gt = pl.nblocks > 0 ? target / pl.nblocks : 0;
  if (gt > V3_MAX_SLOTS) gt = V3_MAX_SLOTS;
  const int gt_cap = kn.gt_cap >= 0 ? kn.gt_cap : pl.chunks / kn.gt_cap_div;
  if (gt > gt_cap) gt = gt_cap;
  if (partials && gt < 1) gt = 1;
  if (partials && kn.xcd > 0 && pl.nblocks > 0) {
    int want = target / pl.nblocks;  // (not capped by gt_cap: the XCD-aware split trades chunks per workgroup for L2 locality)
    if (want > kn.xcd) want = kn.xcd;
    const bool short_k = pl.chunks <= 32 && pl.nblocks <= 32 && kn.short_k_xcd > 0 && kn.short_k_xcd < want;
    if (short_k) want = kn.short_k_xcd;
    if (want > 8) want = 8;
    if (want > pl.chunks) want = pl.chunks;
    int g2 = 1, sh = 0;
    while (g2 * 2 <= want) { g2 *= 2; ++sh; }
    // Same box, us per launch (round 3): o 3584 x 3584 8.75 -> 8.11 (8 slices instead of 5), down 3584 x 18944 18.5 -> 18.2 (8
    // instead of 9), FETCH_SIZE of the 128-row launches 13.4 -> 9.6 MB on average (down 28.2 -> 19.9: the activation matrix
    // crosses the fabric once instead of once per XCD); q|k|v 4608 x 3584 would drop to 4 slices = 144 workgroups and is
    // SLOWER (9.19 -> 9.45): taken only when the launch still fills >= 85 % of the CUs.
    if ((pl.nblocks * g2) % 8 == 0 && (short_k || pl.nblocks * g2 * 100 >= target * kn.fill)) {
      gt = g2;
      pl.xcd_shift = sh;
    }
  }
  if ((lead > 0 || partials) && gt >= (partials ? 1 : 2) && pl.chunks - lead >= gt) {
    pl.gt = gt;
    pl.glead = lead;
    pl.gbase = (pl.chunks - lead) / gt;
    pl.grem = (pl.chunks - lead) % gt;
    pl.upw = pl.gbase + lead;
    pl.grid = pl.nblocks * gt;
    pl.slots = gt;
    return pl;
  }
  int upw = (pl.total_units + target - 1) / target;
  // a tile has at most (chunks - 2) / upw + 2 contributors
  const int min_upw = (pl.chunks + (V3_MAX_SLOTS - 2) - 1) / (V3_MAX_SLOTS - 2);
  if (upw < min_upw) upw = min_upw;
  if (upw < 2) upw = 2;
  pl.upw = upw;
  pl.grid = (pl.total_units + upw - 1) / upw;
  int slots = (pl.chunks - 1) / upw + 2;
  if (slots > V3_MAX_SLOTS) slots = V3_MAX_SLOTS;
  pl.slots = slots;
  return pl;
}

static bool v3_shape_ok(int64_t m, int64_t n, int64_t k) {
  if (!((m >= 1) && (m <= V3_BM) && (n >= V3_BN) && (n % V3_BN == 0) && (k >= V3_CK) && (k % V3_CK == 0))) return false;
  return n * k / 2 < (1ll << 31) && n * (k / V3_CK) * 8 < (1ll << 31);  // 32-bit buffer offsets
}

extern "C" int ll_w4a16_prepacked_supported(int64_t m, int64_t n, int64_t k, int group_size) {
  if (group_size <= 0 || group_size % 128 != 0 || k % group_size != 0) return 0;
  const int gdiv = group_size / 128;
  return v3_shape_ok(m, n, k) && ((gdiv & (gdiv - 1)) == 0) ? 1 : 0;
}

// Split-K partial mode (epilogue 2 of ll_w4a16_matmul_prepacked): number of fp32 partials [S][M][N] the launch
// writes; 0 when the shape has too many tiles for the tile-group split (use the ordinary epilogue then).
static int v3_partials_count(int64_t m, int64_t n, int64_t k, int group_size) {
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size)) return 0;
  const V3Plan pl = v3_plan(n, k, 1, true);
  return pl.gt >= 1 && pl.grid == pl.nblocks * pl.gt ? pl.gt : 0;
}
// `epilogue` = the value the launch will be given: bits 8-9 (a forced unit-loop tile width, tests / tuning) keep the launch off the
// short-stream engine (gemm_short.hip), whose k-split differs from the unit loop's.
extern "C" int ll_w4a16_partials_count_ex(int64_t m, int64_t n, int64_t k, int group_size, int epilogue) {
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size)) return 0;
  if (!((epilogue >> 8) & 3)) {
    const int s = ss_partials_slices(m, n, k, group_size);
    if (s > 0) return s;
  }
  return v3_partials_count(m, n, k, group_size);
}
extern "C" int ll_w4a16_partials_count(int64_t m, int64_t n, int64_t k, int group_size) {
  return ll_w4a16_partials_count_ex(m, n, k, group_size, 2);
}

// The launch plan of ll_w4a16_matmul_prepacked for (n, k, epilogue) as 16 ints -- host-side introspection for tests and
// DESIGN_NOTEBOOK.md (no device work): [0] grid, [1] 128-row blocks per tile, [2] tiles, [3] chunks, [4] slab slots, [5] gt, [6] gbase,
// [7] grem, [8] glead, [9] xcd_shift, [10] upw, [11..14] 0 (reserved), [15] compute units assumed.
extern "C" int ll_w4a16_v3_plan(int64_t m, int64_t n, int64_t k, int group_size, int epilogue, int32_t* out16) {
  if (!out16) return LL_ERR_ARG;
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size)) return LL_ERR_SHAPE;
  const bool partials = (epilogue & 3) == 2;
  if (partials && !v3_partials_count(m, n, k, group_size)) return LL_ERR_SHAPE;
  const V3Plan pl = v3_plan(n, k, (epilogue >> 8) & 3, partials);
  const int v[16] = {pl.grid, pl.nf, pl.nblocks, pl.chunks, pl.slots, pl.gt, pl.gbase, pl.grem, pl.glead,
                     (partials && pl.gt >= 1 && pl.grid == pl.nblocks * pl.gt) ? pl.xcd_shift : -1, pl.upw, 0, 0,
                     0, 0, v3_knobs().wgs > 0 ? v3_knobs().wgs : v3_num_cus()};
  for (int i = 0; i < 16; ++i) out16[i] = v[i];
  return LL_OK;
}

extern "C" int ll_w4a16_v3_workspace(int64_t m, int64_t n, int64_t k, int64_t* floats, int64_t* ints) {
  if (floats) *floats = 0;
  if (ints) *ints = 0;
  if (!v3_shape_ok(m, n, k)) return LL_OK;
  if (floats) *floats = 0;
  for (int nf = 0; nf <= 2; ++nf) {  // the scratch fits whichever tile width a call asks for
    const V3Plan pl = v3_plan(n, k, nf);
    const int64_t f = (int64_t)pl.nblocks * pl.nf * pl.slots * V3_SLAB;
    if (floats && f > *floats) *floats = f;
  }
  if (ints) *ints = n / V3_BN * 8 + 1;  // merge counters + the error word
  return LL_OK;
}

static int v3_launch(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias, int64_t m, int64_t n,
                     int64_t k, int group_size, int64_t x_stride_m, float* workspace, int32_t* counters, int epilogue,
                     void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || group_size <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size) || x_stride_m % 8 != 0 || ((epilogue & 1) && (n & 1))) return LL_ERR_SHAPE;
  if (!out || !x || !wpacked || !spacked || !workspace || !counters) return LL_ERR_ARG;
  if (!ll_aligned16(x) || !ll_aligned16(wpacked) || !ll_aligned16(spacked)) return LL_ERR_ARG;
  if ((m - 1) * x_stride_m * 2 + k * 2 >= (1ll << 31)) return LL_ERR_SHAPE;
  const bool partials = (epilogue & 3) == 2;
  if (partials && (bias || !v3_partials_count(m, n, k, group_size))) return LL_ERR_SHAPE;
  // round 6: split-K partial launches whose stream is a few tens of KB per CU run on the short-stream engine (gemm_short.hip)
  if (partials && !((epilogue >> 8) & 3) && ss_partials_slices(m, n, k, group_size) > 0)
    return ss_launch(out, x, wpacked, spacked, m, n, k, group_size, x_stride_m, stream);
  // ... and narrow FINISHED outputs (the reference-shaped layer's q / k|v / o calls, the fused gate|up of a TP >= 4 shard) on its
  // all-of-K form (gemm_short_full.hip): the sums meet inside the workgroup, no slabs, no counters
  if (!partials && !v4_wants(m, n, k, group_size, epilogue) && sf_wants(m, n, k, group_size, epilogue))
    return sf_launch(out, x, wpacked, spacked, bias, m, n, k, group_size, x_stride_m, epilogue, stream);
  const V3Plan pl = v3_plan(n, k, (epilogue >> 8) & 3, partials);
  // round 5: the row-group engine (gemm_w4_v4.hip) takes the launch when its plan serves the shape -- same layouts, same planes
  if (v4_wants(m, n, k, group_size, epilogue))
    return v4_launch(out, x, wpacked, spacked, bias, m, n, k, group_size, x_stride_m, epilogue & 3, partials ? pl.gt : 1, stream);
  epilogue &= 3;
  V3Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.wp = wpacked; p.sp = spacked; p.bias = (const uint16_t*)bias;
  p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m;
  p.x_cstride = V3_CK * 2;
  p.w_bytes = (uint32_t)(n * k / 2);
  p.s_bytes = (uint32_t)(n * (k / group_size) * 8);
  p.x_bytes = (uint32_t)((m - 1) * x_stride_m * 2 + k * 2);
  p.nblocks = pl.nblocks; p.chunks = pl.chunks; p.total_units = pl.total_units; p.upw = pl.upw; p.slots = pl.slots;
  p.gt = pl.gt; p.gbase = pl.gbase; p.grem = pl.grem; p.glead = pl.glead;
  p.epi = epilogue;
  auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };
  // umulhi(x, ceil(2^32 / d)) == x / d for x * d < 2^32 (x: unit / workgroup indices); d == 1 is handled by the guard below
  if ((int64_t)pl.total_units * (pl.chunks > pl.upw ? pl.chunks : pl.upw) >= (1ll << 31)) return LL_ERR_SHAPE;
  p.err_idx = (int)(n / V3_BN * 8);
  p.xcd_shift = (partials && pl.gt >= 1 && pl.grid == pl.nblocks * pl.gt) ? pl.xcd_shift : -1;
  p.chunks_magic = magic(pl.chunks);
  p.gt_magic = magic(pl.gt);
  p.upw_magic = magic(pl.upw);
  V3_DEBUG_SET(p)
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V3Lds<1>::BYTES);
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V3Lds<1>::BYTES);
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, V3Lds<2>::BYTES);
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, V3Lds<2>::BYTES);
    attr_set[dev] = true;
  }
  const dim3 grid((unsigned)pl.grid);
  if (pl.nf == 2) {
    if (m <= 32) wgemm3_kernel<1, 2><<<grid, V3_THREADS, V3Lds<2>::BYTES, st>>>(p);
    else wgemm3_kernel<2, 2><<<grid, V3_THREADS, V3Lds<2>::BYTES, st>>>(p);
  } else {
    if (m <= 32) wgemm3_kernel<1, 1><<<grid, V3_THREADS, V3Lds<1>::BYTES, st>>>(p);
    else wgemm3_kernel<2, 1><<<grid, V3_THREADS, V3Lds<1>::BYTES, st>>>(p);
  }
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_w4a16_matmul_prepacked(void* out, const void* x, const void* wpacked, const void* spacked,
                                         const void* bias, int64_t m, int64_t n, int64_t k, int group_size,
                                         int64_t x_stride_m, float* workspace, int32_t* counters, int epilogue,
                                         void* stream) {
  return v3_launch(out, x, wpacked, spacked, bias, m, n, k, group_size, x_stride_m, workspace, counters, epilogue, stream);
}
