// W4A16 dequant-GEMM, decode engine, third generation (M <= 64, group_size = 128 * 2^j) over weights
// PRE-PACKED at load time (ll_w4a16_pack_weights) -- the load-time layout slot the reference reserves in
// lite_llama/models/quantization/_layout/__init__.py:1-6.  Semantics: lite_llama/kernels/quantization/w4a16.py:28-207
// (out[m, n] = sum_k x[m, k] * (nib(n, k) - z[n, k/g]) * s[n, k/g] (+ bias), fp32 accumulation, fp16 out).
//
// What round 1 measured on the second-generation engine (gemm_w4_v2.hip) and what this one
// changes:
//   * the loop was paced by the consumer waves at 0.8 us per 8-KB unit (matrix-pipe floor 0.30), with four
//     memory waves (loaders: global -> registers -> swizzled LDS ring; producers: x through v_perm) sharing
//     the consumers' SIMDs and one workgroup-wide barrier tying twelve waves together -> here there are NO
//     memory roles: the weights are stored in the order the MFMA wants them, so every consumer wave streams
//     ITS OWN weight words (one contiguous KB per wave and unit) and scale pairs straight from global memory
//     into a register ring V3_R units deep, and stages its eighth of the activation tile; nothing is
//     permuted on the way (the nibble order inside a word is arranged by the packer so that the dequantised
//     pairs come out in natural k order);
//   * 5 us of prologue (unit table in LDS, a barrier, then a 112-KB request burst per CU) -> the unit sequence
//     is three scalar segments (tail, full tiles, head) walked by two scalar cursors; the first loads leave
//     before anything else happens;
//   * every global access is a raw buffer load: one scalar offset per unit, fixed per-lane offsets, rows
//     >= M and units past the end of the range read as zeros without touching memory (null descriptor).
// Kept from v2: the unit (128 weight rows x 128 k), the 4 row groups x 2 k-halves consumer shape, the exact
// nibble unpack + fp16 affine map (13 VALU per 8 weights), stream-K with static tile ownership (tail segment
// first, head segment last; contributors park partials in slabs with write-through stores and post a
// counter, the owner merges with coherent loads; waits only point at lower-numbered workgroups) and the
// tile-group split with an owner lead for shapes with few tiles.  New in the merge: after the k-half
// reduction each of the two waves of a row group finishes ONE 32-row half of the batch (flush, merge and
// epilogue are split two ways instead of idling the k-half-1 wave).
#include <stdlib.h>

#include "common.h"

#define V3_BN 128
#define V3_BM 64
#define V3_CK 128
#define V3_THREADS 512
#define V3_R 5  // register ring: a unit's operands are requested V3_R units before they are used
#define V3_MAX_SLOTS 12
#define V3_FRAG 1024                  // floats of one (row group, batch half) partial: 16 per lane
#define V3_SLAB (V3_BN * V3_BM)       // floats per (tile, contributor)
#define V3_A_ROW 272                  // padded x-tile row (conflict-free ds_read_b128 across 16 rows)
#define V3_A_TILE (V3_BM * V3_A_ROW)  // 17408
#define V3_OFF_A 0                    // two x tiles
#define V3_OFF_R (2 * V3_A_TILE)      // k-half exchange: 4 row groups x 2 batch halves x 4 KB
#define V3_LDS_BYTES (V3_OFF_R + 8 * 4096)
#define V3_SPIN_LIMIT (1 << 18)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
  uint32_t w_bytes, s_bytes, x_bytes;
  int nblocks, chunks, total_units, upw, slots;
  int gt, gbase, grem, glead;  // tile-group split, see v3_plan
  int gshift;                  // log2(group_size / 128)
  int epi;                     // 0: out[m, n];  1: rows are (gate_j, up_j) pairs -> out[m, n/2] = swiglu
};

__device__ __forceinline__ uint32_t v3_pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t v3_pk_fma(uint32_t a, uint32_t b, uint32_t c) {
  f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b),
                                      __builtin_bit_cast(f16x2, c));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t v3_and_or(uint32_t w, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask), "v"(magic));
  return r;
}
// One packed word -> eight fp16 weights in natural k order (the packer stores nibble 2i of a k-octet at
// position i and nibble 2i+1 at position 4+i).  Exact unpack: (w & 0x000F000F) | 0x6400 = (1024 + q_i,
// 1024 + q_{4+i}); the offset is removed exactly, then ONE fp16 fma with (s, -z*s): the same arithmetic as
// gemm_w4_v2.hip / gemm_wq.hip.  Error against the reference's fp16(fp32((q - z) * s)): <= 3 fp16 ulps of the
// weight (roundings of s, of z*s and of the fma), stated in DESIGN.md.
__device__ __forceinline__ f16x8 v3_dequant(uint32_t w, uint32_t s, uint32_t nzs, uint32_t magic) {
  const uint32_t w2 = w >> 8;
  uint32_t a = v3_and_or(w, 0x000F000Fu, magic);
  uint32_t b = v3_and_or(w, 0x00F000F0u, magic);
  uint32_t c = v3_and_or(w2, 0x000F000Fu, magic);
  uint32_t d = v3_and_or(w2, 0x00F000F0u, magic);
  a = v3_pk_add(a, 0xE400E400u);
  b = v3_pk_fma(b, 0x2C002C00u, 0xD400D400u);
  c = v3_pk_add(c, 0xE400E400u);
  d = v3_pk_fma(d, 0x2C002C00u, 0xD400D400u);
  u32x4 o;
  o.x = v3_pk_fma(a, s, nzs);
  o.y = v3_pk_fma(b, s, nzs);
  o.z = v3_pk_fma(c, s, nzs);
  o.w = v3_pk_fma(d, s, nzs);
  return __builtin_bit_cast(f16x8, o);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t v3_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

// Scalar cursor over the workgroup's unit sequence: up to three segments (tail of the last tile, the
// full tiles, head of the first tile), each a run of consecutive chunks that wraps into the next tile.
struct V3Cur {
  int t, c, left, seg;
};

template <int MT>
__global__ __launch_bounds__(V3_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgemm3_kernel(const V3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = p.chunks;
  int ub, ue;
  if (p.gt) {
    const int gtile = (int)blockIdx.x / p.gt, j = (int)blockIdx.x - gtile * p.gt;
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
  const int tA = ub / chunks, cA = ub - tA * chunks;
  const int tZ = (ue - 1) / chunks, cZ = (ue - 1) - tZ * chunks;
  int LT = 0, LH = cnt;  // a range inside one tile runs as a single "head" segment
  if (tA != tZ) {
    LT = (cZ != chunks - 1) ? cZ + 1 : 0;
    LH = (cA != 0) ? chunks - cA : 0;
  }
  const int NF = cnt - LT - LH;
  const int tF = tA + ((LH > 0 && tA != tZ) ? 1 : 0);
  // segments in execution order, empty ones squeezed out (selects only: a runtime-indexed array would live in scratch)
  const bool hasT = LT > 0, hasF = NF > 0;
  const int sg_t0 = hasT ? tZ : (hasF ? tF : tA), sg_c0 = hasT ? 0 : (hasF ? 0 : cA), sg_n0 = hasT ? LT : (hasF ? NF : LH);
  const int sg_t1 = (hasT && hasF) ? tF : tA, sg_c1 = (hasT && hasF) ? 0 : cA;
  const int sg_n1 = hasT ? (hasF ? NF : LH) : (hasF ? LH : 0);
  const int sg_t2 = tA, sg_c2 = cA, sg_n2 = (hasT && hasF) ? LH : 0;
  auto cur_advance = [&](V3Cur& cu) {  // selects only (scalar ALU, no branches between the loads)
    const int c1 = cu.c + 1;
    const bool wrap = c1 == chunks;
    const int left = cu.left - 1;
    const bool nextseg = left == 0;
    const int s = cu.seg + (nextseg ? 1 : 0);
    const int nt = s == 1 ? sg_t1 : sg_t2, nc = s == 1 ? sg_c1 : (s == 2 ? sg_c2 : 0);
    const int nn = s == 1 ? sg_n1 : (s == 2 ? sg_n2 : 0);
    cu.t = nextseg ? nt : (wrap ? cu.t + 1 : cu.t);
    cu.c = nextseg ? nc : (wrap ? 0 : c1);
    cu.left = nextseg ? (nn == 0 ? 0x40000000 : nn) : left;  // past the end: stays there
    cu.seg = s;
  };

  const int ng = wv & 3, kh = wv >> 2;
  const int nl = lane & 31, h = lane >> 5;

  // fixed per-lane byte offsets of the three streams
  const int w_voff = (wv * 64 + lane) * 16;
  const int s_voff = (ng * 32 + nl) * 8;
  int x_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t row = (wv * 2 + j) * 4 + (lane >> 4);
    x_voff[j] = row < p.m ? (int)(row * p.x_stride * 2 + (lane & 15) * 16) : 0x7FFFFFF0;  // rows >= M read as zeros
  }
  const int x_lds = ((wv * 2) * 4 + (lane >> 4)) * V3_A_ROW + (lane & 15) * 16;  // + j * 4 rows

  // register ring
  u32x4 rW[V3_R];
  u32x2 rS[V3_R];
  u32x4 rX[V3_R][2];

  V3Cur lc{sg_t0, sg_c0, sg_n0, 0};  // load cursor
  int lleft = cnt;                         // units the load cursor still has to request
  auto load_x = [&](int i) {
    const bool live = lleft > 0;
    const __amdgpu_buffer_rsrc_t rx = v3_rsrc(p.x, live ? p.x_bytes : 0u);
    const int so = live ? lc.c * (V3_CK * 2) : 0;
    rX[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_voff[0], so, 0);
    rX[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_voff[1], so, 0);
  };
  auto load_w = [&](int i) {
    const bool live = lleft > 0;
    const __amdgpu_buffer_rsrc_t rw = v3_rsrc(p.wp, live ? p.w_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs = v3_rsrc(p.sp, live ? p.s_bytes : 0u);
    const int wo = live ? (lc.t * chunks + lc.c) * (V3_BN * V3_CK / 2) : 0;
    const int so = live ? ((lc.c >> p.gshift) * (int)p.n + lc.t * V3_BN) * 8 : 0;
    rW[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff, wo, 0);
    rS[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, s_voff, so, 0);
    cur_advance(lc);
    lleft -= 1;
  };
  auto stage_x = [&](int i, int slot) {
    unsigned char* dst = lds + V3_OFF_A + slot * V3_A_TILE + x_lds;
    *reinterpret_cast<u32x4*>(dst) = rX[i][0];
    *reinterpret_cast<u32x4*>(dst + 4 * V3_A_ROW) = rX[i][1];
  };

  // ------------------------------ prologue: V3_R units requested at once ------------------------------ //
#pragma unroll
  for (int i = 0; i < V3_R; ++i) {
    load_x(i);
    load_w(i);
  }

  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  f32x16 acc0, acc1;  // two values, not an array: a wave-uniform choice between them must stay a select
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  };
  zero_acc();
  const int aoff = nl * V3_A_ROW + kh * 128 + h * 64;

  // contribution counter of a parked partial, posted once its write-through stores have landed
  int32_t* pend_ctr = nullptr;
  int pend_val = 0;
  auto post_pending = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(pend_ctr, pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_ctr = nullptr;
  };

  // Finish one 32-row batch half `mt` of row group ng for tile t, whose chunks [c_lo, c_hi] this workgroup
  // has just summed into v (k-halves already added).
  auto flush = [&](f32x16& v, int mt, int t, int c_lo, int c_hi) {
    const int w0 = p.gt ? t * p.gt : (int)((uint32_t)(t * chunks) / (uint32_t)p.upw);  // first contributor of the tile
    const int slot = (int)blockIdx.x - w0;
    int32_t* ctr = &p.counters[(t * 4 + ng) * 2 + mt];
    if (c_hi != chunks - 1) {
      // contributor: park the partial in this workgroup's slab (counter follows, see post_pending)
      float* ws = p.workspace + ((((int64_t)t * p.slots + slot) * 4 + ng) * 2 + mt) * V3_FRAG;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        float* dst = ws + (g * 64 + lane) * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(q) : "memory");
      }
      pend_ctr = ctr;
      pend_val = c_hi - c_lo + 1;
      return;
    }
    if (c_lo != 0) {
      // owner: chunks [0, c_lo) were summed by the `slot` lower-numbered contributors
      for (int spin = 0; spin < V3_SPIN_LIMIT; ++spin) {
        const int seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(seen) >= c_lo) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // slabs were written through (sc1) before their counter: coherent (sc1) loads, no acquire fence.
      // Four slabs in flight per round trip; the tail of the last round is masked to +0.
      for (int sl = 0; sl < slot; sl += 4) {
        i32x4 va[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int sq = sl + q < slot ? sl + q : sl;
          const float* src = p.workspace + ((((int64_t)t * p.slots + sq) * 4 + ng) * 2 + mt) * V3_FRAG + lane * 4;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(va[q][g]) : "v"(src + g * 256) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(va[q][g]));  // uses stay below the wait
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int keep = sl + q < slot ? -1 : 0;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * g + e] += __int_as_float(va[q][g][e] & keep);
        }
      }
    }
    const int64_t mrow = nl + mt * 32;
    if (mrow >= p.m) return;
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t nn = (int64_t)t * V3_BN + ng * 32 + 8 * g + 4 * h;
      uint16_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f = v[4 * g + e];
        if (has_bias) f += f16_bits_to_f32(p.bias[nn + e]);
        o[e] = f32_to_f16_bits(f);
      }
      if (p.epi) {
        // weight rows 2j / 2j+1 are gate_j / up_j: both land in this lane.  Same arithmetic as the
        // stand-alone kernels: the two GEMM outputs rounded to fp16, then silu(g) * u in fp32.
        const float g0 = f16_bits_to_f32(o[0]), u0 = f16_bits_to_f32(o[1]);
        const float g1 = f16_bits_to_f32(o[2]), u1 = f16_bits_to_f32(o[3]);
        const uint32_t s0 = f32_to_f16_bits(g0 * ll_sigmoidf(g0) * u0);
        const uint32_t s1 = f32_to_f16_bits(g1 * ll_sigmoidf(g1) * u1);
        *reinterpret_cast<uint32_t*>(p.out + mrow * (p.n >> 1) + (nn >> 1)) = s0 | (s1 << 16);
      } else {
        uint2 pk;
        pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2*>(p.out + mrow * p.n + nn) = pk;
      }
    }
  };

  // End of a tile segment: the two k-halves of a row group exchange one batch half each through LDS, so
  // that wave kh ends up with the complete sums of batch half kh (MT = 2), or kh = 0 with everything (MT = 1).
  auto segment_end = [&](int t, int c_lo, int c_hi) {
    float* red = reinterpret_cast<float*>(lds + V3_OFF_R) + ng * (2 * V3_FRAG);
    auto put = [&](const f32x16& a, int half) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(red + half * V3_FRAG + (g * 64 + lane) * 4) =
            f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    };
    auto get_add = [&](f32x16& a, int half) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(red + half * V3_FRAG + (g * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * g + e] += q[e];
      }
    };
    if constexpr (MT == 2) {
      f32x16 give, v;  // one copy of the exchange / flush code for both waves of the row group
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        give[r] = kh == 0 ? acc1[r] : acc0[r];
        v[r] = kh == 0 ? acc0[r] : acc1[r];
      }
      put(give, 1 - kh);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      get_add(v, kh);
      flush(v, kh, t, c_lo, c_hi);
    } else {
      if (kh == 1) put(acc0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kh == 0) {
        get_add(acc0, 0);
        flush(acc0, 0, t, c_lo, c_hi);
      }
    }
    zero_acc();
  };

  auto compute = [&](int i, int slot) {
    const unsigned char* ab = lds + V3_OFF_A + slot * V3_A_TILE + aoff;
    f16x8 af[4][MT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[s][mt] = *reinterpret_cast<const f16x8*>(ab + mt * 32 * V3_A_ROW + s * 16);
    __builtin_amdgcn_sched_barrier(0);  // all fragment reads in flight before the first dequant (hipcc sinks them to their use otherwise)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t word = s == 0 ? rW[i].x : s == 1 ? rW[i].y : s == 2 ? rW[i].z : rW[i].w;
      const f16x8 wfrag = v3_dequant(word, rS[i].x, rS[i].y, magic);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, af[s][0], acc0, 0, 0, 0);
      if constexpr (MT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, af[s][1], acc1, 0, 0, 0);
    }
  };

  V3Cur cc{sg_t0, sg_c0, sg_n0, 0};  // compute cursor
  int seg_lo = cc.c;
  int slot = 0;
  int done = 0;
  stage_x(0, 0);
  __syncthreads();

  // One unit.  Ring set I holds unit `done`; set (I + 1) % R the next unit, whose x eighth is staged now
  // into the other LDS tile (every wave left that tile at the barrier that ended the previous unit).
#define V3_STEP(I)                                                                          \
  {                                                                                         \
    stage_x(((I) + 1) % V3_R, slot ^ 1);                                                    \
    load_x(I);                                                                              \
    compute(I, slot);                                                                       \
    load_w(I);                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                         \
    if (pend_ctr) post_pending(); /* the previous segment's slab stores are a unit old */   \
    const bool se_ = (cc.c == chunks - 1) | (cc.left == 1);                                 \
    if (se_) segment_end(cc.t, seg_lo, cc.c);                                               \
    cur_advance(cc);                                                                        \
    if (se_) seg_lo = cc.c;                                                                 \
    slot ^= 1;                                                                              \
    ++done;                                                                                 \
  }
  for (int it = cnt / V3_R; it > 0; --it) {
    V3_STEP(0)
    V3_STEP(1)
    V3_STEP(2)
    V3_STEP(3)
    V3_STEP(4)
  }
  if (done < cnt) V3_STEP(0)
  if (done < cnt) V3_STEP(1)
  if (done < cnt) V3_STEP(2)
  if (done < cnt) V3_STEP(3)
#undef V3_STEP
  if (pend_ctr) post_pending();
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

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct V3Plan {
  int nblocks, chunks, total_units, upw, grid, slots;
  int gt, gbase, grem, glead;
};

struct V3Knobs {
  int wgs = 0, lead = 4, gt_cap_div = 5, gt_cap = -1;
  V3Knobs() {
    if (const char* e = getenv("LL_GEMM3_WGS")) wgs = atoi(e);
    if (const char* e = getenv("LL_GEMM3_LEAD")) lead = atoi(e);
    if (const char* e = getenv("LL_GEMM3_GT")) gt_cap = atoi(e);
    if (const char* e = getenv("LL_GEMM3_GTDIV")) gt_cap_div = atoi(e) > 0 ? atoi(e) : 5;
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

static V3Plan v3_plan(int64_t n, int64_t k) {
  const V3Knobs& kn = v3_knobs();
  V3Plan pl;
  pl.nblocks = (int)(n / V3_BN);
  pl.chunks = (int)(k / V3_CK);
  pl.total_units = pl.nblocks * pl.chunks;
  const int target = kn.wgs > 0 ? kn.wgs : v3_num_cus();  // one persistent 8-wave workgroup per CU
  pl.gt = pl.gbase = pl.grem = pl.glead = 0;
  // Few tiles: every tile is shared by gt workgroups; the last one (the owner) runs `lead` chunks longer
  // than the contributors, so their slabs and counters have landed by the time it merges.
  const int lead = kn.lead;
  int gt = pl.nblocks > 0 ? target / pl.nblocks : 0;
  if (gt > V3_MAX_SLOTS) gt = V3_MAX_SLOTS;
  const int gt_cap = kn.gt_cap >= 0 ? kn.gt_cap : pl.chunks / kn.gt_cap_div;
  if (gt > gt_cap) gt = gt_cap;
  if (lead > 0 && gt >= 2 && pl.chunks - lead >= gt) {
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

extern "C" int ll_w4a16_v3_workspace(int64_t m, int64_t n, int64_t k, int64_t* floats, int64_t* ints) {
  if (floats) *floats = 0;
  if (ints) *ints = 0;
  if (!v3_shape_ok(m, n, k)) return LL_OK;
  const V3Plan pl = v3_plan(n, k);
  if (floats) *floats = (int64_t)pl.nblocks * pl.slots * V3_SLAB;
  if (ints) *ints = (int64_t)pl.nblocks * 8;
  return LL_OK;
}

extern "C" int ll_w4a16_matmul_prepacked(void* out, const void* x, const void* wpacked, const void* spacked,
                                         const void* bias, int64_t m, int64_t n, int64_t k, int group_size,
                                         int64_t x_stride_m, float* workspace, int32_t* counters, int epilogue,
                                         void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || group_size <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  if (!ll_w4a16_prepacked_supported(m, n, k, group_size) || x_stride_m % 8 != 0 || (epilogue && (n & 1))) return LL_ERR_SHAPE;
  if (!out || !x || !wpacked || !spacked || !workspace || !counters) return LL_ERR_ARG;
  if (!ll_aligned16(x) || !ll_aligned16(wpacked) || !ll_aligned16(spacked)) return LL_ERR_ARG;
  if ((m - 1) * x_stride_m * 2 + k * 2 >= (1ll << 31)) return LL_ERR_SHAPE;
  const V3Plan pl = v3_plan(n, k);
  V3Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.wp = wpacked; p.sp = spacked; p.bias = (const uint16_t*)bias;
  p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m;
  p.w_bytes = (uint32_t)(n * k / 2);
  p.s_bytes = (uint32_t)(n * (k / group_size) * 8);
  p.x_bytes = (uint32_t)((m - 1) * x_stride_m * 2 + k * 2);
  p.nblocks = pl.nblocks; p.chunks = pl.chunks; p.total_units = pl.total_units; p.upw = pl.upw; p.slots = pl.slots;
  p.gt = pl.gt; p.gbase = pl.gbase; p.grem = pl.grem; p.glead = pl.glead;
  p.epi = epilogue;
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)wgemm3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES);
    attr_set[dev] = true;
  }
  if (m <= 32)
    wgemm3_kernel<1><<<dim3((unsigned)pl.grid), V3_THREADS, V3_LDS_BYTES, st>>>(p);
  else
    wgemm3_kernel<2><<<dim3((unsigned)pl.grid), V3_THREADS, V3_LDS_BYTES, st>>>(p);
  return LL_LAUNCH_CHECK();
}
