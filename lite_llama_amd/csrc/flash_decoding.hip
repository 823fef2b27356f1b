// flash_decoding (a5) -- reference lite_llama/kernels/flashdecoding.py:23-380.
//
// MI355X design (NOT the reference's one-program-per-q-head layout):
//   * stage 1: one wave per (batch row, KV head, 128-token partition) serves ALL
//     hq/hkv query heads of that KV head, so every K/V byte is read from HBM once
//     (the reference re-reads it hq/hkv times).
//   * S^T = K.Q^T on MFMA 16x16x32 (tokens x heads): K fragments are loaded straight
//     from the token-attention pool in fragment layout (64 contiguous bytes per lane
//     per gathered row), q-heads sit in the 16 MFMA columns, so the online softmax
//     is lane-local per head plus two cross-row shuffles.
//   * O^T = V^T.P^T on MFMA: the S^T accumulator registers ARE the P^T B-operand
//     (no data movement); V rows are staged once through LDS to be read k(token)-major.
//   * stage 2: log-sum-exp merge of the per-partition partials (fp32 scratch).
// Numerics follow the reference: fp32 scores/accumulators, plain exp, plain scale.
#include <stdlib.h>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Tokens per partition: the reference's PARTITION_SIZE (flashdecoding.py:343).  -DFD_PART=64 builds the
// 64-token variant (twice the waves in flight, half the gather chain per wave): measured SLOWER on the
// headline shape (3.82 vs 3.58 ms/step, same box) -- twice the partials, a second merge round trip.
#ifndef FD_PART
#define FD_PART 128
#endif
#ifndef FD_MERGE_PB4
#define FD_MERGE_PB4 5
#endif
#define FD_LSE_PAD 32  // floats per log-sum-exp record in the one-launch form (one cache line each)

struct alignas(16) Q4 {
  uint32_t x, y, z, w;
};

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const Q4& a, const Q4& b, f32x4 c) {
  if constexpr (DT == LL_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)from_f32<DT>(lo) | ((uint32_t)from_f32<DT>(hi) << 16);
}

__device__ __forceinline__ int64_t fd_load_idx(const void* p, int64_t i, int w) {
  return w == LL_I32 ? (int64_t)((const int32_t*)p)[i] : ((const int64_t*)p)[i];
}

// ROPE (decode step, one launch for rope + KV scatter + attention): q arrives UN-rotated and the new
// token's K/V in ``kv_new`` ([batch, 2*hkv, D] rows, K heads first).  Every wave rotates its q
// fragments in registers (the rotation partner d +- D/2 sits in the same lane: fragment s +- NS/2);
// the wave of the partition that holds the new token rotates k_new and writes K and V into pool row
// select_index[b] before it gathers (nobody else reads that row).  Element arithmetic is the rope
// kernel's: fp32 products of the 16-bit values, rounded once.  Needs D >= 64, one head group.
struct FdRope {
  const uint16_t* kv_new;
  int64_t kv_rs;              // kv_new row stride (elements)
  const uint16_t* cos_t;      // [max_pos, >= D/2] tables, row stride cs_rs
  const uint16_t* sin_t;
  int64_t cs_rs;
  const int64_t* positions;   // [batch]
  const void* sel;            // select_index [batch]
  int sel_w;
  uint16_t* pool_k;           // same storage as kc / vc, writable
  uint16_t* pool_v;
  // q / k_new / v_new left by the fused q|k|v projection as fp32 split-K partials [qs][batch][row_w] (GROUPED only):
  // the workgroup adds them up (+ bias, one fp16 rounding: the value the projection would have stored) into LDS
  const float* qp;            // nullptr: q / kv_new are finished fp16 tensors
  int qs;                     // number of partials (<= FD_QS_MAX)
  int64_t qp_plane, qp_row;   // elements per partial, per batch row
  const uint16_t* qbias;      // [row_w] or nullptr
  float k_scale, v_scale;     // fp8 KV cache (KV8): stored value * scale = K / V value
  // QKN (Qwen3): per-head RMSNorm of q and of the new K row BEFORE the rotation -- weights [D] of the q dtype
  const uint16_t* qnw;
  const uint16_t* knw;
  float nrm_eps;
  // QI32 (smoothquant q|k|v): the planes at qp are EXACT int32 sums; value = fp16(((float)sum * q_as[row]) * q_ws[col] (+ bias))
  const float* q_as;          // [batch] per-token activation scales
  const float* q_ws;          // [row_w] per-channel weight scales
};
#define FD_QS_MAX 8

// grid = (nparts, hkv * head_groups, batch), block = 64 (one wave)
// FUSE: the wave that finishes the LAST non-empty partition of its (row, KV head group) also does the
// log-sum-exp merge and writes ``out`` (no second launch: a kernel boundary costs ~4.7 us in the
// decode graph, the merge itself ~1 us).  Partials are written through (sc1), a relaxed agent-scope
// counter orders them, the merging wave reads them with coherent (sc1) loads and zeroes the counter.
// GS: upper bound (4, 8 or 16) of the query heads per KV head handled by this wave (merge layout).
// GROUPED (decode contexts of up to FD_GROUP_MAX partitions): ONE workgroup per (row, KV head group), one wave per
// partition; the partials meet in LDS and wave 0 merges them -- no partial stores, no counter, no coherent
// re-load: the merge tail shrinks from two global round trips (~5 us at batch 64 x ctx 512) to a barrier.
#ifndef FD_GROUP_MAX
#define FD_GROUP_MAX 8
#endif
// KV8 (extension): the pool holds OCP e4m3 bytes (ll_update_kv_buffer_fp8); fragments are gathered as 8-byte pieces and
// widened to fp16 in registers (v_cvt_scalef32_pk_f16_fp8, exact), k_scale rides on the softmax scale and v_scale on
// the final normalisation -- half the K/V bytes, the same MFMA arithmetic.  fp16 queries only, no in-kernel rope.
struct alignas(8) Q2 {
  uint32_t x, y;
};
template <bool KV8>
struct FdKv {
  using Reg = Q4;
  using Elem = uint16_t;
  static __device__ __forceinline__ Q4 frag(const Q4& r) { return r; }
};
template <>
struct FdKv<true> {
  using Reg = Q2;
  using Elem = uint8_t;
  static __device__ __forceinline__ Q4 frag(const Q2& r) {
    Q4 o;
    o.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)r.x, 1.0f, false));
    o.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)r.x, 1.0f, true));
    o.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)r.y, 1.0f, false));
    o.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)r.y, 1.0f, true));
    return o;
  }
};

// K/V gathers.  FD_NT: non-temporal policy for the WHOLE-LINE gathers (FD_FULL_LINE below) -- same box, batch 64, us per
// launch at a context of 512 / 640 / 1024: 17.1 / 21.6 / 30.5 against 18.8 / 23.5 / 33.3 with the default policy (3.93 / 4.4
// TB/s instead of 3.57 / 4.05).  The half-line gathers (64 bytes of 16 rows per instruction: the fp8 pool, head size 64)
// keep the default policy: nt measured SLOWER there (20.5 vs 18.7 us) -- a line's second half is served by the L1 that nt
// bypasses.
#ifndef FD_NT
#define FD_NT 1
#endif
typedef unsigned fd_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned fd_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT = false>
__device__ __forceinline__ Q4 fd_gather(const Q4* p) {
  if constexpr (NT) return __builtin_bit_cast(Q4, __builtin_nontemporal_load(reinterpret_cast<const fd_u32x4*>(p)));
  else return *p;
}
template <bool NT = false>
__device__ __forceinline__ Q2 fd_gather(const Q2* p) {
  if constexpr (NT) return __builtin_bit_cast(Q2, __builtin_nontemporal_load(reinterpret_cast<const fd_u32x2*>(p)));
  else return *p;
}

// FD_FULL_LINE (head size 128, 16-bit pool): every gather instruction covers WHOLE 128-byte lines -- lane (t, c) fetches row
// t & 7 of an 8-row group, 16-byte piece c + 4 (t >> 3) of one line -- instead of 64-byte halves of 16 rows; lanes t and t ^ 8
// then trade the piece the other one's MFMA fragment needs (one row_ror:8 DPP move per register).  V rows go to LDS
// anyway: only their store addresses change.
#ifndef FD_FULL_LINE
#define FD_FULL_LINE 1
#endif
__device__ __forceinline__ Q4 fd_ror8(const Q4& v) {
  Q4 o;
  o.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.x, 0x128, 0xf, 0xf, true);
  o.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.y, 0x128, 0xf, 0xf, true);
  o.z = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.z, 0x128, 0xf, 0xf, true);
  o.w = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.w, 0x128, 0xf, 0xf, true);
  return o;
}
__device__ __forceinline__ Q4 fd_sel(bool p, const Q4& a, const Q4& b) {
  return Q4{p ? a.x : b.x, p ? a.y : b.y, p ? a.z : b.z, p ? a.w : b.w};
}
// raw[g * 2 + l] = (row g * 8 + (t & 7), line l, piece c + 4 hi)  ->  frag[s] = (row t, bytes s * 64 + c * 16)
__device__ __forceinline__ void fd_fix_fragments(Q4 (&r)[4], bool hi) {
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const Q4 keep = fd_sel(hi, r[2 + l], r[l]);
    const Q4 recv = fd_ror8(fd_sel(hi, r[l], r[2 + l]));
    r[l] = keep;      // parked; re-ordered below
    r[2 + l] = recv;
  }
  // keep[l] = piece (hi ? 4 + c : c) of line l, recv[l] = the other one: s = 2 l + (piece >= 4)
  const Q4 k0 = r[0], k1 = r[1], v0 = r[2], v1 = r[3];
  r[0] = fd_sel(hi, v0, k0);
  r[1] = fd_sel(hi, k0, v0);
  r[2] = fd_sel(hi, v1, k1);
  r[3] = fd_sel(hi, k1, v1);
}

// QKN: the per-head RMSNorm Qwen3 applies to q (and to the new K row) between the projection and the rotation --
// reference lite_llama/models/qwen3.py q_norm / k_norm -> kernels/skip_rmsnorm.py.  Same values as ll_skip_rmsnorm on the
// [rows * heads, 128] view, bit for bit: thread tr of skip_rmsnorm_cached<TPR = 16> owns elements tr * 8 .. + 8, which is
// fragment s, piece c of this lane (tr = 4 s + c); its xor butterfly over tr (8, 4, 2, 1) is fragments s ^ 2, s ^ 1, then
// lanes ^ 32, ^ 16 -- reproduced in that order.  Head size 128 only.
template <int DT>
__device__ __forceinline__ float fd_ssq8(const Q4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  float p = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = to_f32<DT>((uint16_t)w[i]), hi = to_f32<DT>((uint16_t)(w[i] >> 16));
    p += lo * lo / 128.f;
    p += hi * hi / 128.f;
  }
  return p;
}
template <int DT>
__device__ __forceinline__ Q4 fd_scale8(const Q4& v, const Q4& wt, float rr) {
  const uint32_t x[4] = {v.x, v.y, v.z, v.w}, w[4] = {wt.x, wt.y, wt.z, wt.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint16_t lo = mul_storage<DT>(from_f32<DT>(to_f32<DT>((uint16_t)x[i]) * rr), (uint16_t)w[i]);
    const uint16_t hi = mul_storage<DT>(from_f32<DT>(to_f32<DT>((uint16_t)(x[i] >> 16)) * rr), (uint16_t)(w[i] >> 16));
    o[i] = (uint32_t)lo | ((uint32_t)hi << 16);
  }
  return Q4{o[0], o[1], o[2], o[3]};
}
template <int DT>
__device__ __forceinline__ void fd_head_norm(Q4 (&qf)[4], const Q4 (&wf)[4], float eps) {
  const float p0 = fd_ssq8<DT>(qf[0]), p1 = fd_ssq8<DT>(qf[1]), p2 = fd_ssq8<DT>(qf[2]), p3 = fd_ssq8<DT>(qf[3]);
  float v = (p0 + p2) + (p1 + p3);
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  const float rr = 1.0f / sqrtf(v + eps);
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = fd_scale8<DT>(qf[s], wf[s], rr);
}

template <int DT, int D, bool FUSE, bool ROPE, int GS, bool GROUPED, bool KV8 = false, bool QKN = false, bool QI32 = false>
__global__ __launch_bounds__(GROUPED ? 64 * FD_GROUP_MAX : 64) void fd_stage1(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int32_t* __restrict__ table, const void* __restrict__ b_req_idx,
    const void* __restrict__ b_seq_len, float* __restrict__ mid_o, float* __restrict__ mid_lse,
    int hq, int hkv, int nparts, float scale, int64_t q_sb, int64_t q_sh, int64_t k_st, int64_t k_sh,
    int64_t v_st, int64_t v_sh, int64_t t_sb, int req_w, int seq_w, uint16_t* __restrict__ out, int64_t o_sb,
    int64_t o_sh, int32_t* __restrict__ counters, FdRope rp) {
  constexpr bool FDFL = FD_FULL_LINE && !KV8 && D == 128;  // whole-line gathers (see fd_fix_fragments)
  constexpr int NS = D / 32;      // MFMA k-steps over the head dim
  constexpr int NT = D / 16;      // output d-tiles
  constexpr int VSTR = D + 8;     // padded LDS row stride (elements)
  static_assert(!KV8 || (DT == LL_F16 && !ROPE), "fp8 KV: fp16 queries, rope and the KV write happen before the launch");
  static_assert(!QKN || (ROPE && D == 128), "q / k head norm: part of the one-launch decode form, head size 128");
  static_assert(!QI32 || (GROUPED && ROPE && DT == LL_F16), "int32 q|k|v planes (smoothquant): the grouped partial-input form, fp16");
  using KVR = typename FdKv<KV8>::Reg;
  using KVE = typename FdKv<KV8>::Elem;
  const KVE* kcE = reinterpret_cast<const KVE*>(kc);
  const KVE* vcE = reinterpret_cast<const KVE*>(vc);
  static_assert(!GROUPED || (FUSE && 32 * VSTR * 2 >= (16 * D + 16) * 4), "a wave's V tile doubles as its partial record");
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = GROUPED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  uint16_t* lds_v = lds_all + wave * (32 * VSTR);  // this wave's V staging tile
  const int t = lane & 15, c = lane >> 4;
  const int part = GROUPED ? wave : (int)blockIdx.x;
  const int kvh = blockIdx.y % hkv;
  const int hg = blockIdx.y / hkv;
  const int b = blockIdx.z;
  const int groups = hq / hkv;
  const int hl = hg * 16 + t;            // head within the KV group (MFMA column)
  const bool head_ok = hl < groups;
  const int head = kvh * groups + hl;

  // ---- q | k_new | v_new of this (row, KV head) from split-K partials: requested first (they depend on nothing
  //      but the block index), summed and published to LDS once the dependent index loads have been issued ----
  uint16_t* q_lds = lds_all + (GROUPED ? (int)(blockDim.x >> 6) : 1) * (32 * VSTR);
  constexpr bool QPART_OK = GROUPED && ROPE;
  const bool qpart = QPART_OK && rp.qp != nullptr;
  f32x4 qacc[QPART_OK ? FD_QS_MAX : 1][QPART_OK ? 2 : 1];
  int qe[2] = {0, 0};
  int64_t qcol[2] = {0, 0};
  bool qok[2] = {false, false};
  uint2 qbv[2] = {uint2{0, 0}, uint2{0, 0}};  // the 4 bias values of a slot, fetched with the partials (one 8-byte load)
  f32x4 qws[QI32 ? 2 : 1];                     // QI32: the slot's 4 weight scales
  float qas = 0.f;                             // QI32: this row's activation scale
  if constexpr (QI32) qas = rp.q_as[b];
  if constexpr (QPART_OK) {
    if (qpart) {
      const int nq = groups * D;                // this KV head's query values, then its new K row, then its new V row
      const int nslot = (nq + 2 * D) / 4;
#pragma unroll
      for (int it = 0; it < 2; ++it) {          // up to 2 float4 slots per lane (>= 2 waves: 128 lanes x 2 >= 288 slots at GQA 7)
        const int slot = (int)threadIdx.x + it * (int)blockDim.x;
        qok[it] = slot < nslot;
        const int e = qok[it] ? slot * 4 : 0;
        qe[it] = e;
        qcol[it] = e < nq ? (int64_t)kvh * nq + e
                          : (e < nq + D ? (int64_t)hq * D + (int64_t)kvh * D + (e - nq)
                                        : (int64_t)hq * D + (int64_t)hkv * D + (int64_t)kvh * D + (e - nq - D));
#pragma unroll
        for (int sl = 0; sl < FD_QS_MAX; ++sl) {
          const int ss = sl < rp.qs ? sl : 0;   // clamped; dropped below
          qacc[sl][it] = *reinterpret_cast<const f32x4*>(rp.qp + ss * rp.qp_plane + b * rp.qp_row + qcol[it]);
        }
        // unconditional (a pointer select, not a branch: a load inside a branch is waited for with vmcnt(0), and the
        // per-element form of this cost eight dependent round trips in the launch's ramp)
        const uint16_t* bsrc = rp.qbias ? rp.qbias + qcol[it] : reinterpret_cast<const uint16_t*>(rp.qp);
        qbv[it] = *reinterpret_cast<const uint2*>(bsrc);
        if constexpr (QI32) qws[it] = *reinterpret_cast<const f32x4*>(rp.q_ws + qcol[it]);
      }
    }
  }
  auto publish_q = [&]() {
    if constexpr (QPART_OK) {
      if (qpart) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if constexpr (QI32) {
            // exact int32 sums, then (acc.f32 * a_scale[m]) * w_scale[n] (w8a8.py:118-120) -- the product pinned to fp32 before
            // the narrowing below (see w8a8_fused.hip::q8_f32)
            i32x4 ai = {0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < FD_QS_MAX; ++sl)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float bits = qacc[sl][it][e];  // (a copy first: __builtin_bit_cast on a vector ELEMENT reads element 0)
                ai[e] += sl < rp.qs ? __float_as_int(bits) : 0;
              }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = ((float)ai[e] * qas) * qws[it][e];
              asm volatile("" : "+v"(v));
              a[e] = v;
            }
          } else {
#pragma unroll
          for (int sl = 0; sl < FD_QS_MAX; ++sl)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += sl < rp.qs ? qacc[sl][it][e] : 0.f;
          }
          uint16_t o4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t w = e < 2 ? qbv[it].x : qbv[it].y;
            const float bias = to_f32<DT>((uint16_t)(e & 1 ? w >> 16 : w));
            o4[e] = from_f32<DT>(rp.qbias ? a[e] + bias : a[e]);
          }
          if (qok[it])
            *reinterpret_cast<uint2*>(q_lds + qe[it]) =
                uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
        }
        __syncthreads();  // every wave of the workgroup passes here exactly once (empty partitions included)
      }
    }
  };

  // (requested together: the table row address needs req, the partition bounds need seq_len -- one round trip, not two)
  const int64_t req = fd_load_idx(b_req_idx, b, req_w);
  const int64_t seq_len = fd_load_idx(b_seq_len, b, seq_w);
  int64_t start = (int64_t)part * FD_PART;  // (GROUPED: the wave's FIRST partition; it walks part, part + waves, ... below)
  if (start >= seq_len) {  // empty partition stores nothing (flashdecoding.py:141-161)
    publish_q();
    if constexpr (FUSE) {
      // a zero-length row has no merging wave: its first partition writes the 0/0 the reference's
      // stage 2 computes for it (:264-287)
      if (part == 0 && head_ok) {
        const float nan = __builtin_nanf("");
        for (int dd = c; dd < D; dd += 4) out[b * o_sb + (int64_t)head * o_sh + dd] = from_f32<DT>(nan);
      }
    }
    if constexpr (GROUPED) __syncthreads();  // the workgroup's one barrier (the waves with tokens meet here before the merge)
    return;
  }
  int64_t end = seq_len < start + FD_PART ? seq_len : start + FD_PART;
  const int32_t* trow = table + req * t_sb;
  // Pool rows of the whole partition, requested now (the K/V gathers depend on them; everything up to the first
  // gather overlaps this round trip)
  const int64_t lastt = end - 1;
  const int64_t tk0 = start + lane < lastt ? start + lane : lastt;
  const int64_t tk1 = start + 64 + lane < lastt ? start + 64 + lane : lastt;
  int r0 = trow[tk0];
  int r1 = FD_PART == 128 ? trow[tk1] : r0;
  publish_q();

  // Q^T fragments (MFMA B operand): lane (head t, group c) holds q[head][s*32 + c*8 .. +8] -- the
  // natural k order of MFMA step s, so the 4 lanes of a row read 64 contiguous bytes per instruction
  Q4 qf[NS];
  if (qpart) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      qf[s] = head_ok ? *reinterpret_cast<const Q4*>(q_lds + hl * D + s * 32 + c * 8) : Q4{0, 0, 0, 0};
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (head_ok)
        qf[s] = *reinterpret_cast<const Q4*>(q + b * q_sb + (int64_t)head * q_sh + s * 32 + c * 8);
      else
        qf[s] = Q4{0, 0, 0, 0};
    }
  }

  constexpr int HS = ROPE ? NS / 2 : 1;
  Q4 cf[HS], sf[HS];
  Q4 nwf[QKN ? NS : 1];
  if constexpr (QKN) {
#pragma unroll
    for (int s = 0; s < NS; ++s) nwf[s] = *reinterpret_cast<const Q4*>(rp.qnw + s * 32 + c * 8);
  }
  if constexpr (ROPE) {
    static_assert(!ROPE || NS >= 2, "the rotation partner must be another fragment of the same lane");
    const int64_t crow = rp.positions[b] * rp.cs_rs;
#pragma unroll
    for (int s = 0; s < HS; ++s) {
      cf[s] = *reinterpret_cast<const Q4*>(rp.cos_t + crow + s * 32 + c * 8);
      sf[s] = *reinterpret_cast<const Q4*>(rp.sin_t + crow + s * 32 + c * 8);
    }
    // the new token lives in the last non-empty partition; head group 0 owns the write
    // (clamped to the launched partitions: a stale / too small max length then still stores the row)
    const int wpart = (int)((seq_len - 1) / FD_PART) < nparts - 1 ? (int)((seq_len - 1) / FD_PART) : nparts - 1;
    // (GROUPED with more partitions than waves: the wave that WILL gather partition wpart -- it is its only reader -- writes now)
    if (hg == 0 && (GROUPED ? wave == wpart % (int)(blockDim.x >> 6) : part == wpart)) {
      const int64_t dstrow = fd_load_idx(rp.sel, b, rp.sel_w);
      const uint16_t* kn = qpart ? q_lds + groups * D : rp.kv_new + b * rp.kv_rs + (int64_t)kvh * D;
      const uint16_t* vn = qpart ? q_lds + groups * D + D : rp.kv_new + b * rp.kv_rs + (int64_t)(hkv + kvh) * D;
      if constexpr (QKN) {
        // every lane takes part (the butterfly needs whole groups of 8 lanes); lanes >= D / 16 repeat lane & 7's work
        const int j = (lane & (D / 16 - 1)) * 8;
        U16x8 k1 = *reinterpret_cast<const U16x8*>(kn + j), k2 = *reinterpret_cast<const U16x8*>(kn + D / 2 + j);
        const U16x8 cv = *reinterpret_cast<const U16x8*>(rp.cos_t + crow + j);
        const U16x8 sv = *reinterpret_cast<const U16x8*>(rp.sin_t + crow + j);
        const U16x8 w1 = *reinterpret_cast<const U16x8*>(rp.knw + j), w2 = *reinterpret_cast<const U16x8*>(rp.knw + D / 2 + j);
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = to_f32<DT>(k1.v[e]), bb = to_f32<DT>(k2.v[e]);
          pa += a * a / 128.f;
          pb += bb * bb / 128.f;
        }
        float var = pa + pb;  // skip_rmsnorm_cached's butterfly: thread tr ^ 8 is this lane's other half, then lanes ^ 4, ^ 2, ^ 1
        var += __shfl_xor(var, 4, 64);
        var += __shfl_xor(var, 2, 64);
        var += __shfl_xor(var, 1, 64);
        const float rr = 1.0f / sqrtf(var + rp.nrm_eps);
        U16x8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = to_f32<DT>(mul_storage<DT>(from_f32<DT>(to_f32<DT>(k1.v[e]) * rr), w1.v[e]));
          const float bb = to_f32<DT>(mul_storage<DT>(from_f32<DT>(to_f32<DT>(k2.v[e]) * rr), w2.v[e]));
          const float cc = to_f32<DT>(cv.v[e]), ss = to_f32<DT>(sv.v[e]);
          o1.v[e] = from_f32<DT>(a * cc - bb * ss);
          o2.v[e] = from_f32<DT>(bb * cc + a * ss);
        }
        if (lane < D / 16) {
          uint16_t* pk = rp.pool_k + dstrow * k_st + (int64_t)kvh * k_sh;
          *reinterpret_cast<U16x8*>(pk + j) = o1;
          *reinterpret_cast<U16x8*>(pk + D / 2 + j) = o2;
        } else if (lane < D / 16 + D / 8) {
          const int jv = (lane - D / 16) * 8;
          *reinterpret_cast<U16x8*>(rp.pool_v + dstrow * v_st + (int64_t)kvh * v_sh + jv) =
              *reinterpret_cast<const U16x8*>(vn + jv);
        }
      } else
      if (lane < D / 16) {
        const int j = lane * 8;
        const U16x8 k1 = *reinterpret_cast<const U16x8*>(kn + j), k2 = *reinterpret_cast<const U16x8*>(kn + D / 2 + j);
        const U16x8 cv = *reinterpret_cast<const U16x8*>(rp.cos_t + crow + j);
        const U16x8 sv = *reinterpret_cast<const U16x8*>(rp.sin_t + crow + j);
        U16x8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = to_f32<DT>(k1.v[e]), bb = to_f32<DT>(k2.v[e]);
          const float cc = to_f32<DT>(cv.v[e]), ss = to_f32<DT>(sv.v[e]);
          o1.v[e] = from_f32<DT>(a * cc - bb * ss);
          o2.v[e] = from_f32<DT>(bb * cc + a * ss);
        }
        uint16_t* pk = rp.pool_k + dstrow * k_st + (int64_t)kvh * k_sh;
        *reinterpret_cast<U16x8*>(pk + j) = o1;
        *reinterpret_cast<U16x8*>(pk + D / 2 + j) = o2;
      } else if (lane < D / 16 + D / 8) {
        const int j = (lane - D / 16) * 8;
        *reinterpret_cast<U16x8*>(rp.pool_v + dstrow * v_st + (int64_t)kvh * v_sh + j) =
            *reinterpret_cast<const U16x8*>(vn + j);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row is in L2 before this wave gathers it
    }
  }

  // scores are kept in the log2 domain (one v_exp_f32 per probability; plain expf costs ~15 VALU each): the
  // running maximum m_i is in log2 units and is converted back where the log-sum-exp record is written
  const float scale2 = scale * 1.44269504088896340736f * (KV8 ? rp.k_scale : 1.0f);
  const uint16_t* vtr = lds_v + (4 * c + (t >> 2)) * VSTR + (t & 3) * 4;
  float m_i = -INFINITY, d_i = 0.f;
  f32x4 ot[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) ot[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (r0 / r1 above: lane l holds the pool rows of tokens start+l and start+64+l, clamped to the last valid token, so
  // every later gather reads a valid row; the tiles pick theirs with a lane shuffle instead of a dependent table
  // load per tile.)
  static_assert(FD_PART == 128 || FD_PART == 64, "the tile schedule below is written for 2 or 4 tiles of 32 tokens");

  // ---- gather of tile TI (32 tokens): lane (t, c) fetches rows TI*32+t and TI*32+16+t, d-range
  //      {s*32 + c*8 .. +8}, straight into MFMA fragment layout (per instruction the 4 lanes of a row
  //      read 64 contiguous bytes: 16 sectors per wave-load instead of 64 partial ones).  Unconditional (clamped rows): hipcc
  //      drains vmcnt(0) around any branch that contains a load.
#define FD_LOAD(S, TI)                                                                         \
  {                                                                                            \
    const int rsrc_ = (TI) < 2 ? r0 : r1;                                                      \
    if constexpr (FDFL) {                                                                      \
      const int tl_ = t & 7, pc_ = (c + 4 * (t >> 3)) * 8;                                     \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                          \
        const int64_t rowA_ = __shfl(rsrc_, ((TI) & 1) * 32 + g * 8 + tl_, 64);                \
        const int64_t rowB_ = __shfl(rsrc_, ((TI) & 1) * 32 + 16 + g * 8 + tl_, 64);           \
        _Pragma("unroll") for (int l = 0; l < 2; ++l) {                                        \
          ka##S[g * 2 + l] = fd_gather<FD_NT != 0>(reinterpret_cast<const KVR*>(kcE + rowA_ * k_st + (int64_t)kvh * k_sh + l * 64 + pc_)); \
          kb##S[g * 2 + l] = fd_gather<FD_NT != 0>(reinterpret_cast<const KVR*>(kcE + rowB_ * k_st + (int64_t)kvh * k_sh + l * 64 + pc_)); \
        }                                                                                      \
      }                                                                                        \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                          \
        const int64_t rowA_ = __shfl(rsrc_, ((TI) & 1) * 32 + g * 8 + tl_, 64);                \
        const int64_t rowB_ = __shfl(rsrc_, ((TI) & 1) * 32 + 16 + g * 8 + tl_, 64);           \
        _Pragma("unroll") for (int l = 0; l < 2; ++l) {                                        \
          va##S[g * 2 + l] = fd_gather<FD_NT != 0>(reinterpret_cast<const KVR*>(vcE + rowA_ * v_st + (int64_t)kvh * v_sh + l * 64 + pc_)); \
          vb##S[g * 2 + l] = fd_gather<FD_NT != 0>(reinterpret_cast<const KVR*>(vcE + rowB_ * v_st + (int64_t)kvh * v_sh + l * 64 + pc_)); \
        }                                                                                      \
      }                                                                                        \
    } else {                                                                                   \
    const int64_t rowA_ = __shfl(rsrc_, ((TI) & 1) * 32 + t, 64);                              \
    const int64_t rowB_ = __shfl(rsrc_, ((TI) & 1) * 32 + 16 + t, 64);                         \
    const KVE* kA_ = kcE + rowA_ * k_st + (int64_t)kvh * k_sh + c * 8;                        \
    const KVE* kB_ = kcE + rowB_ * k_st + (int64_t)kvh * k_sh + c * 8;                        \
    const KVE* vA_ = vcE + rowA_ * v_st + (int64_t)kvh * v_sh + c * 8;                        \
    const KVE* vB_ = vcE + rowB_ * v_st + (int64_t)kvh * v_sh + c * 8;                        \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) ka##S[s] = fd_gather(reinterpret_cast<const KVR*>(kA_ + s * 32)); \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) kb##S[s] = fd_gather(reinterpret_cast<const KVR*>(kB_ + s * 32)); \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) va##S[s] = fd_gather(reinterpret_cast<const KVR*>(vA_ + s * 32)); \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) vb##S[s] = fd_gather(reinterpret_cast<const KVR*>(vB_ + s * 32)); \
    }                                                                                          \
  }

  // The V tile belongs to ONE wave: its LDS operations execute in order, so within a multi-wave (GROUPED) workgroup a
  // drained counter orders write -> read; the single-wave form keeps the workgroup barrier.
#define FD_WAVE_SYNC()                                                    \
  do {                                                                    \
    if constexpr (GROUPED) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    else __syncthreads();                                                 \
  } while (0)
  // ---- one 32-token tile: S^T, online softmax, O^T += V^T P^T ----
#define FD_COMPUTE(S, TI)                                                                      \
  {                                                                                            \
    const int64_t pos0 = start + (TI) * 32;                                                    \
    const bool okA = pos0 + t < end, okB = pos0 + 16 + t < end;                                \
    if constexpr (FDFL) {                                                                      \
      fd_fix_fragments(ka##S, t >= 8);                                                         \
      fd_fix_fragments(kb##S, t >= 8);                                                         \
    }                                                                                          \
    /* S^T tiles: rows = tokens (4c+r), cols = heads (t) */                                    \
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};                                \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) sa = mfma16<DT>(FdKv<KV8>::frag(ka##S[s]), qf[s], sa); \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) sb = mfma16<DT>(FdKv<KV8>::frag(kb##S[s]), qf[s], sb); \
    float sc[8];                                                                               \
    float mx = -INFINITY;                                                                      \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
      const bool va_ok = pos0 + 4 * c + r < end;                                               \
      const bool vb_ok = pos0 + 16 + 4 * c + r < end;                                          \
      sc[r] = va_ok ? sa[r] * scale2 : -INFINITY;                                              \
      sc[4 + r] = vb_ok ? sb[r] * scale2 : -INFINITY;                                          \
      mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));                                                 \
    }                                                                                          \
    mx = rows_max(mx);                                                                         \
    const float m_new = fmaxf(m_i, mx); /* finite: token pos0 is always valid */               \
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);                                   \
    float p[8];                                                                                \
    float ps = 0.f;                                                                            \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
      p[j] = __builtin_amdgcn_exp2f(sc[j] - m_new);                                            \
      ps += p[j];                                                                              \
    }                                                                                          \
    ps = rows_sum(ps);                                                                         \
    d_i = d_i * alpha + ps;                                                                    \
    m_i = m_new;                                                                               \
    /* P^T fragment (MFMA B operand): k-slot j <-> token 16*(j>>2) + 4c + (j&3) */             \
    Q4 pf;                                                                                     \
    pf.x = pack2<DT>(p[0], p[1]);                                                              \
    pf.y = pack2<DT>(p[2], p[3]);                                                              \
    pf.z = pack2<DT>(p[4], p[5]);                                                              \
    pf.w = pack2<DT>(p[6], p[7]);                                                              \
    /* stage V rows through LDS (zero rows past the end: garbage could be NaN) */              \
    FD_WAVE_SYNC(); /* previous tile's reads done */                                           \
    if constexpr (FDFL) { /* raw registers: (row g * 8 + (t & 7), line l, piece c + 4 (t >> 3)) */ \
      const int tl_ = t & 7, pc_ = (c + 4 * (t >> 3)) * 8;                                     \
      const bool full_ = pos0 + 32 <= end;                                                     \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                          \
        const bool okA_ = full_ || pos0 + g * 8 + tl_ < end, okB_ = full_ || pos0 + 16 + g * 8 + tl_ < end; \
        _Pragma("unroll") for (int l = 0; l < 2; ++l) {                                        \
          *reinterpret_cast<Q4*>(&lds_v[(g * 8 + tl_) * VSTR + l * 64 + pc_]) = okA_ ? va##S[g * 2 + l] : Q4{0, 0, 0, 0};      \
          *reinterpret_cast<Q4*>(&lds_v[(16 + g * 8 + tl_) * VSTR + l * 64 + pc_]) = okB_ ? vb##S[g * 2 + l] : Q4{0, 0, 0, 0}; \
        }                                                                                      \
      }                                                                                        \
    } else                                                                                     \
    if (pos0 + 32 <= end) { /* whole tile inside the context (wave-uniform): no masking */     \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                         \
        *reinterpret_cast<Q4*>(&lds_v[t * VSTR + s * 32 + c * 8]) = FdKv<KV8>::frag(va##S[s]); \
        *reinterpret_cast<Q4*>(&lds_v[(16 + t) * VSTR + s * 32 + c * 8]) = FdKv<KV8>::frag(vb##S[s]); \
      }                                                                                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                         \
        *reinterpret_cast<Q4*>(&lds_v[t * VSTR + s * 32 + c * 8]) = okA ? FdKv<KV8>::frag(va##S[s]) : Q4{0, 0, 0, 0};        \
        *reinterpret_cast<Q4*>(&lds_v[(16 + t) * VSTR + s * 32 + c * 8]) = okB ? FdKv<KV8>::frag(vb##S[s]) : Q4{0, 0, 0, 0}; \
      }                                                                                        \
    }                                                                                          \
    FD_WAVE_SYNC();                                                                            \
    /* O^T[d][head] += V^T[d][tok] . P^T[tok][head] */                                         \
    _Pragma("unroll") for (int dt = 0; dt < NT; ++dt) {                                        \
      /* V^T fragment: lane (d = t, group c) needs V[tok][dt*16 + t] for tok = 4c..4c+3 and 16+4c..+3.  The   \
         transposing LDS read (benchmarks/probes/tr_probe.hip) treats the 16 lanes' 8-byte pieces as a 4 x 16  \
         matrix -- lanes 4r..4r+3 supply row r, four columns each -- and hands lane t column t: lane t points \
         at row 4c + (t >> 2), columns 4 (t & 3)..+3 of the d-tile */                                          \
      const s16x4 tlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                \
          (__attribute__((address_space(3))) s16x4*)(vtr + dt * 16));                          \
      const s16x4 thi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                \
          (__attribute__((address_space(3))) s16x4*)(vtr + 16 * VSTR + dt * 16));             \
      Q4 vf;                                                                                   \
      vf.x = (uint32_t)(uint16_t)tlo.x | ((uint32_t)(uint16_t)tlo.y << 16);                    \
      vf.y = (uint32_t)(uint16_t)tlo.z | ((uint32_t)(uint16_t)tlo.w << 16);                    \
      vf.z = (uint32_t)(uint16_t)thi.x | ((uint32_t)(uint16_t)thi.y << 16);                    \
      vf.w = (uint32_t)(uint16_t)thi.z | ((uint32_t)(uint16_t)thi.w << 16);                    \
      f32x4 acc = ot[dt];                                                                      \
      acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;                      \
      ot[dt] = mfma16<DT>(vf, pf, acc);                                                        \
    }                                                                                          \
  }

#if defined(FD_ABLATE) && (FD_ABLATE & 1)  // debug: loads only -- the registers are consumed, nothing is computed
#undef FD_COMPUTE
#define FD_COMPUTE(S, TI)                                                                                        \
  _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                               \
    asm volatile("" ::"v"(__builtin_bit_cast(i32x4, ka##S[s])), "v"(__builtin_bit_cast(i32x4, kb##S[s])),          \
                 "v"(__builtin_bit_cast(i32x4, va##S[s])), "v"(__builtin_bit_cast(i32x4, vb##S[s])));             \
  }
#endif
#if defined(FD_ABLATE) && (FD_ABLATE & 2)  // debug: arithmetic only -- no K / V gathers
#undef FD_LOAD
#define FD_LOAD(S, TI)                                                                                           \
  _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                               \
    ka##S[s] = kb##S[s] = va##S[s] = vb##S[s] = Q4{(uint32_t)(TI) * 0x3c003c00u + (uint32_t)r0, (uint32_t)r1 & 0x3fff3fffu, 0x3c003c00u, (uint32_t)s}; \
  }
#endif
  // two register sets: tile i+1 is in flight while tile i is multiplied
  KVR kaA[NS], kbA[NS], vaA[NS], vbA[NS], kaB[NS], kbB[NS], vaB[NS], vbB[NS];
  FD_LOAD(A, 0)
  FD_LOAD(B, 1)
  if constexpr (QKN) {
    fd_head_norm<DT>(qf, nwf, rp.nrm_eps);
    if (!head_ok) {
#pragma unroll
      for (int s = 0; s < NS; ++s) qf[s] = Q4{0, 0, 0, 0};
    }
  }
  if constexpr (ROPE) {
    // (after the first gathers are in flight: the rotation only has to precede the first S^T)
#pragma unroll
    for (int s = 0; s < HS; ++s) {
      const uint32_t* x1 = reinterpret_cast<const uint32_t*>(&qf[s]);
      const uint32_t* x2 = reinterpret_cast<const uint32_t*>(&qf[s + HS]);
      const uint32_t* cw = reinterpret_cast<const uint32_t*>(&cf[s]);
      const uint32_t* sw = reinterpret_cast<const uint32_t*>(&sf[s]);
      uint32_t r1[4], r2[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint16_t lo1, hi1, lo2, hi2;
        {
          const float a = to_f32<DT>((uint16_t)x1[w]), bb = to_f32<DT>((uint16_t)x2[w]);
          const float cc = to_f32<DT>((uint16_t)cw[w]), ss = to_f32<DT>((uint16_t)sw[w]);
          lo1 = from_f32<DT>(a * cc - bb * ss);
          lo2 = from_f32<DT>(bb * cc + a * ss);
        }
        {
          const float a = to_f32<DT>((uint16_t)(x1[w] >> 16)), bb = to_f32<DT>((uint16_t)(x2[w] >> 16));
          const float cc = to_f32<DT>((uint16_t)(cw[w] >> 16)), ss = to_f32<DT>((uint16_t)(sw[w] >> 16));
          hi1 = from_f32<DT>(a * cc - bb * ss);
          hi2 = from_f32<DT>(bb * cc + a * ss);
        }
        r1[w] = (uint32_t)lo1 | ((uint32_t)hi1 << 16);
        r2[w] = (uint32_t)lo2 | ((uint32_t)hi2 << 16);
      }
      qf[s] = Q4{r1[0], r1[1], r1[2], r1[3]};
      qf[s + HS] = Q4{r2[0], r2[1], r2[2], r2[3]};
    }
    }
  FD_COMPUTE(A, 0)
  if constexpr (FD_PART == 128) FD_LOAD(A, 2)
  if (start + 32 < end) FD_COMPUTE(B, 1)
  if constexpr (FD_PART == 128) {
    FD_LOAD(B, 3)
    if (start + 64 < end) FD_COMPUTE(A, 2)
    if (start + 96 < end) FD_COMPUTE(B, 3)
  }
  if constexpr (GROUPED) {
    // Contexts of more partitions than waves (round 6): the wave walks partitions part + W, part + 2 W, ... with the SAME running
    // maximum / denominator / accumulators -- the online softmax does not care where a partition ends -- so the one-workgroup form
    // (partials of the q|k|v projection summed in the prologue, merge through LDS, no global round trip) serves any context.
    const int W_ = (int)(blockDim.x >> 6);
    for (int pn = part + W_; pn < nparts; pn += W_) {
      const int64_t st_ = (int64_t)pn * FD_PART;
      if (st_ >= seq_len) break;
      start = st_;
      end = seq_len < start + FD_PART ? seq_len : start + FD_PART;
      const int64_t last_ = end - 1;
      const int64_t t0_ = start + lane < last_ ? start + lane : last_;
      const int64_t t1_ = start + 64 + lane < last_ ? start + 64 + lane : last_;
      r0 = trow[t0_];
      r1 = FD_PART == 128 ? trow[t1_] : r0;
      FD_LOAD(A, 0)
      FD_LOAD(B, 1)
      FD_COMPUTE(A, 0)
      if constexpr (FD_PART == 128) FD_LOAD(A, 2)
      if (start + 32 < end) FD_COMPUTE(B, 1)
      if constexpr (FD_PART == 128) {
        FD_LOAD(B, 3)
        if (start + 64 < end) FD_COMPUTE(A, 2)
        if (start + 96 < end) FD_COMPUTE(B, 3)
      }
    }
  }
#undef FD_LOAD
#undef FD_COMPUTE
#undef FD_WAVE_SYNC

  if constexpr (GROUPED) {
    int np = (int)((seq_len + FD_PART - 1) / FD_PART);  // records = waves with tokens
    if (np > nparts) np = nparts;
    if (np > (int)(blockDim.x >> 6)) np = (int)(blockDim.x >> 6);
    constexpr int REC = 32 * VSTR / 2;  // floats per wave record: [16 heads][D] normalised partial + [16] log-sum-exp
    if (np > 1 && head_ok) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last tile's V reads are done: the tile becomes the record
      float* rec = reinterpret_cast<float*>(lds_v);
      const float inv = (KV8 ? rp.v_scale : 1.0f) / d_i;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        f32x4 o = ot[dt];
        o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
        *reinterpret_cast<f32x4*>(rec + t * D + dt * 16 + 4 * c) = o;
      }
      if (c == 0) rec[16 * D + t] = m_i * 0.69314718055994530942f + logf(d_i);
    }
    __syncthreads();
    if (wave != 0) return;
    if (np == 1) {
      // single partition: stage 2 computes (0*0 + 1*(o/d)) / (0*0 + 1) -- the same value
      if (!head_ok) return;
      uint16_t* orow = out + b * o_sb + (int64_t)head * o_sh + 4 * c;
      const float inv = (KV8 ? rp.v_scale : 1.0f) / d_i;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        uint16_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = from_f32<DT>(ot[dt][e] * inv);
        *reinterpret_cast<uint2*>(orow + dt * 16) =
            uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
      }
      return;
    }
    // merge the np records in partition order: the recurrence of fd_stage2, element by element (same values as
    // the global-memory merge below and as the two-launch form)
    constexpr int D4 = D / 4;
    constexpr int NSL = (GS * D4 + 63) / 64;
    const int nvalid = groups - hg * 16 < 16 ? groups - hg * 16 : 16;
    uint16_t* out_g = out + b * o_sb + (int64_t)(kvh * groups + hg * 16) * o_sh;
    const float* recs = reinterpret_cast<const float*>(lds_all);
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const int sidx = lane + 64 * j;
      const int hl_s = sidx / D4, d4 = sidx % D4;
      if (hl_s >= nvalid) continue;
      float mm = -INFINITY, dd = 0.f;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int pp = 0; pp < np; ++pp) {
        const float* rec = recs + pp * REC;
        const f32x4 v = *reinterpret_cast<const f32x4*>(rec + hl_s * D + d4 * 4);
        const float l = rec[16 * D + hl_s];
        const float m_new = fmaxf(mm, l);
        const float alpha = expf(mm - m_new), w = expf(l - m_new);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] * alpha + w * v[e];
        dd = dd * alpha + w;
        mm = m_new;
      }
      uint16_t o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = from_f32<DT>(acc[e] / dd);
      *reinterpret_cast<uint2*>(out_g + (int64_t)hl_s * o_sh + d4 * 4) =
          uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
    }
    return;
  }
  if constexpr (!FUSE) {
    if (head_ok) {
      const float inv = (KV8 ? rp.v_scale : 1.0f) / d_i;
      float* mo = mid_o + (((int64_t)b * hq + head) * nparts + part) * D;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        f32x4 o = ot[dt];
        o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
        *reinterpret_cast<f32x4*>(mo + dt * 16 + 4 * c) = o;
      }
      if (c == 0) mid_lse[((int64_t)b * hq + head) * nparts + part] = m_i * 0.69314718055994530942f + logf(d_i);
    }
  } else {
    int np = (int)((seq_len + FD_PART - 1) / FD_PART);
    if (np > nparts) np = nparts;
    const int64_t hrow = ((int64_t)b * hq + (head_ok ? head : 0)) * nparts;
    if (np > 1) {
      if (head_ok) {
        const float inv = (KV8 ? rp.v_scale : 1.0f) / d_i;
        float* mo = mid_o + (hrow + part) * D;
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
          f32x4 o = ot[dt];
          o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mo + dt * 16 + 4 * c), "v"(o) : "memory");
        }
        // one 128-byte line per (head, partition): a line that the merging wave's own XCD has
        // written into (its own lse) could otherwise serve a stale copy of a neighbour's value
        if (c == 0)
          __hip_atomic_store(mid_lse + (hrow + part) * FD_LSE_PAD, m_i * 0.69314718055994530942f + logf(d_i), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partials are out before the counter moves
      int32_t* ctr = counters + (int64_t)b * gridDim.y + blockIdx.y;
      int old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old != np - 1) return;
      if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    if (np == 1) {
      // single partition: stage 2 computes (0*0 + 1*(o/d)) / (0*0 + 1) -- the same value
      if (!head_ok) return;
      uint16_t* orow = out + b * o_sb + (int64_t)head * o_sh + 4 * c;
      const float inv = (KV8 ? rp.v_scale : 1.0f) / d_i;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        uint16_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = from_f32<DT>(ot[dt][e] * inv);
        *reinterpret_cast<uint2*>(orow + dt * 16) =
            uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
      }
      return;
    }
    // Merge the np partials in partition order (the recurrence of fd_stage2, element by element).
    // The work is re-dealt over ALL 64 lanes -- GS heads x D/4 float4 slots, NSL per lane -- instead
    // of the MFMA layout's one head per 4 lanes (7 of 16 head columns busy at GQA 7): fewer
    // registers per partition, so PB partitions' loads are in flight at once and a context of up
    // to PB x 128 tokens merges in ONE memory round trip (these are coherent loads from memory).
    constexpr int D4 = D / 4;
    constexpr int NSL = (GS * D4 + 63) / 64;
    constexpr int PB = NSL <= 2 ? 8 : (NSL <= 4 ? FD_MERGE_PB4 : (NSL <= 8 ? 2 : 1));
    const int nvalid = groups - hg * 16 < 16 ? groups - hg * 16 : 16;
    float mm[NSL], dd[NSL];
    f32x4 acc[NSL];
    int mo_off[NSL];  // float offsets of the slot's record of partition 0 (mid_o / mid_lse) and of its output
    int ml_off[NSL];
    int out_off[NSL];
    bool ok_s[NSL];
    const int64_t hr0 = ((int64_t)b * hq + kvh * groups + hg * 16) * nparts;  // first head of the group
    const float* mo_g = mid_o + hr0 * D;
    const float* ml_g = mid_lse + hr0 * FD_LSE_PAD;
    uint16_t* out_g = out + b * o_sb + (int64_t)(kvh * groups + hg * 16) * o_sh;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const int sidx = lane + 64 * j;
      const int hl_s = sidx / D4, d4 = sidx % D4;
      ok_s[j] = hl_s < nvalid;
      const int hs = ok_s[j] ? hl_s : 0;
      mo_off[j] = hs * nparts * D + d4 * 4;
      ml_off[j] = hs * nparts * FD_LSE_PAD;
      out_off[j] = (int)(hs * o_sh) + d4 * 4;
      mm[j] = -INFINITY;
      dd[j] = 0.f;
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int p0 = 0; p0 < np; p0 += PB) {
      f32x4 v[PB][NSL];
      float l[PB][NSL];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int pp = p0 + u < np ? p0 + u : np - 1;  // clamped; masked below
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          asm volatile("global_load_dword %0, %1, off sc1" : "=v"(l[u][j]) : "v"(ml_g + (ml_off[j] + pp * FD_LSE_PAD)) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u][j]) : "v"(mo_g + (mo_off[j] + pp * D)) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < PB; ++u)
#pragma unroll
        for (int j = 0; j < NSL; ++j) asm volatile("" : "+v"(l[u][j]), "+v"(v[u][j]));  // uses stay below the wait
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const bool live = p0 + u < np;  // wave-uniform
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          const float m_new = fmaxf(mm[j], l[u][j]);
          const float alpha = expf(mm[j] - m_new), w = expf(l[u][j] - m_new);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] = live ? acc[j][e] * alpha + w * v[u][j][e] : acc[j][e];
          dd[j] = live ? dd[j] * alpha + w : dd[j];
          mm[j] = live ? m_new : mm[j];
          __builtin_amdgcn_sched_barrier(0);  // one slot's exp chain at a time: bounded temporaries
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      if (!ok_s[j]) continue;
      uint16_t o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = from_f32<DT>(acc[j][e] / dd[j]);
      *reinterpret_cast<uint2*>(out_g + out_off[j]) =
          uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
    }
  }
}

// grid = (hq, batch), block = D threads
template <int DT>
__global__ void fd_stage2(uint16_t* __restrict__ out, const float* __restrict__ mid_o,
                          const float* __restrict__ mid_lse, const void* __restrict__ b_seq_len,
                          int hq, int d, int nparts, int64_t o_sb, int64_t o_sh, int seq_w) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int64_t seq_len = fd_load_idx(b_seq_len, b, seq_w);
  int np = (int)((seq_len + FD_PART - 1) / FD_PART);
  if (np > nparts) np = nparts;
  const float* mo = mid_o + ((int64_t)b * hq + h) * nparts * d;
  const float* ml = mid_lse + ((int64_t)b * hq + h) * nparts;
  float m_i = -INFINITY, d_i = 0.f, acc = 0.f;
  // partitions in batches of 8: all 16 loads of a batch are issued before the first use (indices
  // clamped, none inside a branch), so a batch costs one memory round trip instead of eight
  // dependent ones; the merge itself stays the sequential recurrence of the reference (:264-287)
  for (int p0 = 0; p0 < np; p0 += 8) {
    float lse[8], mo_v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pp = p0 + u < np ? p0 + u : np - 1;
      lse[u] = ml[pp];
      mo_v[u] = mo[(int64_t)pp * d + threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool live = p0 + u < np;
      const float m_new = fmaxf(m_i, lse[u]);
      const float alpha = expf(m_i - m_new);
      const float w = expf(lse[u] - m_new);
      const float acc_n = acc * alpha + w * mo_v[u];
      const float d_n = d_i * alpha + w;
      acc = live ? acc_n : acc;
      d_i = live ? d_n : d_i;
      m_i = live ? m_new : m_i;
    }
  }
  out[b * o_sb + (int64_t)h * o_sh + threadIdx.x] = from_f32<DT>(acc / d_i);  // 0/0 -> NaN like the reference
}

extern "C" int ll_flash_decoding_num_partitions(int64_t max_len) {
  return (int)((max_len + FD_PART - 1) / FD_PART);
}
// Waves of the one-workgroup-per-(row, KV head group) decode form for contexts of up to max_len tokens, launched as `workgroups`
// = batch x KV heads x head groups workgroups; 0: the form does not apply.  Up to FD_GROUP_MAX partitions: a wave per partition
// (whatever the batch: round 2's measurement).  More (round 6): FD_GROUP_MAX waves that walk the partitions -- taken when the
// launch alone fills half the chip (128 workgroups); smaller batches keep a workgroup per partition + the counter merge, which
// spreads ONE row's context over many CUs.
#define FD_LOOP_MAX 256  // partitions (32 768 tokens)
extern "C" int ll_flash_decoding_group_waves(int64_t max_len, int64_t workgroups) {
  static const bool no_loop = getenv("LL_FD_NO_LOOP") != nullptr;  // A/B knob, read once
  const int64_t np = (max_len + FD_PART - 1) / FD_PART;
  if (np < 2) return 0;
  if (np <= FD_GROUP_MAX) return (int)np;
  return (!no_loop && np <= FD_LOOP_MAX && workgroups >= 128) ? FD_GROUP_MAX : 0;
}

template <int DT>
static int launch_fd(void* out, const void* q, const void* kc, const void* vc, const int32_t* table,
                     const void* req, const void* seq, float* mid_o, float* mid_lse, int batch, int hq,
                     int hkv, int d, int64_t max_len, float scale, int64_t q_sb, int64_t q_sh,
                     int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_sh,
                     int64_t t_sb, int req_w, int seq_w, int32_t* counters, const FdRope* rope, hipStream_t st,
                     bool kv8 = false, float k_scale = 1.0f, float v_scale = 1.0f) {
  const int nparts = ll_flash_decoding_num_partitions(max_len);
  const int groups = hq / hkv;
  const int hgroups = (groups + 15) / 16;
  dim3 grid((unsigned)nparts, (unsigned)(hkv * hgroups), (unsigned)batch);
  // one launch when the caller lends a (zeroed, self-cleaning) counter vector and ``out`` takes 8-byte stores
  const bool fuse = counters != nullptr && (o_sb % 4 == 0) && (o_sh % 4 == 0) && ((uintptr_t)out % 8 == 0);
  if (rope && (!fuse || hgroups != 1 || d < 64)) return LL_ERR_SHAPE;
  FdRope rp = rope ? *rope : FdRope{};
  rp.k_scale = k_scale;
  rp.v_scale = v_scale;
  const bool qkn = rope && (rp.qnw || rp.knw);
  if (qkn && (!rp.qnw || !rp.knw || d != 128 || kv8 || !(rp.nrm_eps >= 0.f))) return LL_ERR_SHAPE;
  const int gw = ll_flash_decoding_group_waves(max_len, (int64_t)batch * hkv * hgroups);  // waves of a grouped workgroup; 0: not grouped
  const bool grouped_ok = fuse && gw >= 2;
  // split-K partial inputs need the grouped form and two float4 slots per lane: (groups + 2) * d / 4 <= 2 * 64 * waves
  static const bool ungrouped_env = getenv("LL_FD_UNGROUPED") != nullptr;  // A/B knob, read once
  if (rope && rp.qp && (!grouped_ok || rp.qs < 1 || rp.qs > FD_QS_MAX || (groups + 2) * d / 4 > 128 * gw || ungrouped_env))
    return LL_ERR_SHAPE;
  // one workgroup per (row, KV head group) with a wave per partition while the context fits FD_GROUP_MAX partitions
  const bool grouped = grouped_ok && !ungrouped_env;
#define LL_FD1X(DD, FU, RO, GG, GR) LL_FD1XK(DD, FU, RO, GG, GR, false, false, false)
#define LL_FD1XK(DD, FU, RO, GG, GR, K8, QN, QI)                                                            \
  {                                                                                                  \
    constexpr int tile_bytes_ = 32 * (DD + 8) * 2;                                                   \
    if (GR) {                                                                                        \
      static bool attr_ = false;                                                                     \
      if (!attr_) {                                                                                  \
        (void)hipFuncSetAttribute((const void*)fd_stage1<DT, DD, FU, RO, GG, GR, K8, QN, QI>,             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, FD_GROUP_MAX * tile_bytes_ + 18 * DD * 2); \
        attr_ = true;                                                                                \
      }                                                                                              \
    }                                                                                                \
    fd_stage1<DT, DD, FU, RO, GG, GR, K8, QN, QI><<<(GR) ? dim3(1, grid.y, grid.z) : grid, (GR) ? 64 * gw : 64, \
                                        ((GR) ? gw : 1) * tile_bytes_ + ((GR) ? 18 * DD * 2 : 0), st>>>( \
        (const uint16_t*)q, (const uint16_t*)kc, (const uint16_t*)vc, table, req, seq, mid_o, mid_lse, hq, hkv, nparts, \
        scale, q_sb, q_sh, k_st, k_sh, v_st, v_sh, t_sb, req_w, seq_w, (uint16_t*)out, o_sb, o_sh, counters, rp); \
  }
#define LL_FD1(DD, FU, RO, GG)                        \
  if (grouped) { if constexpr (FU) LL_FD1X(DD, FU, RO, GG, true) } \
  else LL_FD1X(DD, FU, RO, GG, false)
#define LL_FD1G(DD, RO)                      \
  if (groups <= 4) LL_FD1(DD, true, RO, 4)   \
  else if (groups <= 8) LL_FD1(DD, true, RO, 8) \
  else LL_FD1(DD, true, RO, 16)
#define LL_FD1D(DD)                   \
  if (rope) {                         \
    if constexpr (DD >= 64) { LL_FD1G(DD, true) } \
  } else if (fuse) { LL_FD1G(DD, false) }         \
  else LL_FD1X(DD, false, false, 16, false)
  if (kv8) {
    // fp8 KV cache: fp16 queries, the one-launch (counter / grouped) forms, head sizes 64 and 128
    if constexpr (DT == LL_F16) {
      if (rope || !fuse || (d != 64 && d != 128)) return LL_ERR_SHAPE;
#define LL_FD8(DD, GG)                                   \
  if (grouped) LL_FD1XK(DD, true, false, GG, true, true, false, false) \
  else LL_FD1XK(DD, true, false, GG, false, true, false, false)
#define LL_FD8G(DD)                  \
  if (groups <= 4) { LL_FD8(DD, 4) } \
  else if (groups <= 8) { LL_FD8(DD, 8) } \
  else { LL_FD8(DD, 16) }
      if (d == 64) { LL_FD8G(64) } else { LL_FD8G(128) }
#undef LL_FD8G
#undef LL_FD8
      return LL_LAUNCH_CHECK();
    } else {
      return LL_ERR_DTYPE;
    }
  }
  if (rope && rp.qp && rp.q_as) {
    // smoothquant q|k|v planes (int32 + scales): the grouped partial-input form (checked above), fp16, head size 128, no head norm
    if constexpr (DT == LL_F16) {
      if (!rp.q_ws || d != 128 || qkn) return LL_ERR_SHAPE;
      if (groups <= 4) { LL_FD1XK(128, true, true, 4, true, false, false, true) }
      else if (groups <= 8) { LL_FD1XK(128, true, true, 8, true, false, false, true) }
      else { LL_FD1XK(128, true, true, 16, true, false, false, true) }
      return LL_LAUNCH_CHECK();
    } else {
      return LL_ERR_DTYPE;
    }
  }
  if (qkn) {
    // one-launch decode form with the q / k head norm in front of the rotation (checked above: rope, d == 128)
#define LL_FD1Q(GG)                                                      \
  if (grouped) LL_FD1XK(128, true, true, GG, true, false, true, false)   \
  else LL_FD1XK(128, true, true, GG, false, false, true, false)
    if (groups <= 4) { LL_FD1Q(4) }
    else if (groups <= 8) { LL_FD1Q(8) }
    else { LL_FD1Q(16) }
#undef LL_FD1Q
    return LL_LAUNCH_CHECK();
  }
  switch (d) {
    case 32: LL_FD1D(32); break;
    case 64: LL_FD1D(64); break;
    case 128: LL_FD1D(128); break;
    case 256: LL_FD1D(256); break;
    default: return LL_ERR_SHAPE;
  }
#undef LL_FD1D
#undef LL_FD1G
#undef LL_FD1
#undef LL_FD1XK
#undef LL_FD1X
  if (!fuse)
    fd_stage2<DT><<<dim3((unsigned)hq, (unsigned)batch), d, 0, st>>>((uint16_t*)out, mid_o, mid_lse, seq, hq, d,
                                                                     nparts, o_sb, o_sh, seq_w);
  return LL_LAUNCH_CHECK();
}

static int fd_entry(void* out, const void* q, const void* k_cache, const void* v_cache, const int32_t* table,
                    const void* b_req_idx, const void* b_seq_len, float* mid_o, float* mid_lse, int batch, int hq,
                    int hkv, int d, int64_t max_len, float qk_scale, int64_t q_stride_b, int64_t q_stride_h,
                    int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                    int64_t o_stride_b, int64_t o_stride_h, int64_t table_stride_b, int dtype, int req_width,
                    int seq_width, int32_t* counters, const FdRope* rope, void* stream, bool kv8 = false,
                    float k_scale = 1.0f, float v_scale = 1.0f) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (kv8 && (dtype != LL_F16 || !(k_scale > 0.f) || !(v_scale > 0.f) || !counters)) return kv8 && dtype != LL_F16 ? LL_ERR_DTYPE : LL_ERR_ARG;
  if ((req_width | seq_width) & ~1) return LL_ERR_DTYPE;
  if (batch < 0 || hq <= 0 || hkv <= 0 || hq % hkv != 0 || max_len < 0) return LL_ERR_SHAPE;
  if (batch == 0) return LL_OK;
  // 16-byte fragment loads: every row/head stride and base must be 8-element aligned
  // (fp8 pool: strides count bytes = elements, pieces are 8 bytes)
  if ((q_stride_b | q_stride_h | k_stride_t | k_stride_h | v_stride_t | v_stride_h) % 8 != 0 || !ll_aligned16(q) ||
      (kv8 ? (((uintptr_t)k_cache | (uintptr_t)v_cache) & 7) != 0 : (!ll_aligned16(k_cache) || !ll_aligned16(v_cache))))
    return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16)
    return launch_fd<LL_F16>(out, q, k_cache, v_cache, table, b_req_idx, b_seq_len, mid_o, mid_lse, batch,
                             hq, hkv, d, max_len, qk_scale, q_stride_b, q_stride_h, k_stride_t, k_stride_h,
                             v_stride_t, v_stride_h, o_stride_b, o_stride_h, table_stride_b, req_width,
                             seq_width, counters, rope, st, kv8, k_scale, v_scale);
  return launch_fd<LL_BF16>(out, q, k_cache, v_cache, table, b_req_idx, b_seq_len, mid_o, mid_lse, batch, hq,
                            hkv, d, max_len, qk_scale, q_stride_b, q_stride_h, k_stride_t, k_stride_h,
                            v_stride_t, v_stride_h, o_stride_b, o_stride_h, table_stride_b, req_width,
                            seq_width, counters, rope, st);
}

extern "C" int ll_flash_decoding(void* out, const void* q, const void* k_cache, const void* v_cache,
                                 const int32_t* table, const void* b_req_idx, const void* b_seq_len,
                                 float* mid_o, float* mid_lse, int batch, int hq, int hkv, int d,
                                 int64_t max_len, float qk_scale, int64_t q_stride_b, int64_t q_stride_h,
                                 int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t,
                                 int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                 int64_t table_stride_b, int dtype, int req_width, int seq_width,
                                 int32_t* counters, void* stream) {
  return fd_entry(out, q, k_cache, v_cache, table, b_req_idx, b_seq_len, mid_o, mid_lse, batch, hq, hkv, d, max_len,
                  qk_scale, q_stride_b, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_b,
                  o_stride_h, table_stride_b, dtype, req_width, seq_width, counters, nullptr, stream);
}

// flash_decoding over an fp8 (OCP e4m3) KV pool written by ll_update_kv_buffer_fp8: k_cache / v_cache are byte
// tensors (strides in bytes), stored value * k_scale (v_scale) = K (V).  fp16 queries, head size 64 or 128, counters
// required (one-launch forms only).  Same arithmetic as ll_flash_decoding on the widened values.
extern "C" int ll_flash_decoding_fp8kv(void* out, const void* q, const void* k_cache, const void* v_cache,
                                       const int32_t* table, const void* b_req_idx, const void* b_seq_len,
                                       float* mid_o, float* mid_lse, int batch, int hq, int hkv, int d,
                                       int64_t max_len, float qk_scale, float k_scale, float v_scale, int64_t q_stride_b,
                                       int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t,
                                       int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                       int64_t table_stride_b, int req_width, int seq_width, int32_t* counters,
                                       void* stream) {
  return fd_entry(out, q, k_cache, v_cache, table, b_req_idx, b_seq_len, mid_o, mid_lse, batch, hq, hkv, d, max_len,
                  qk_scale, q_stride_b, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_b,
                  o_stride_h, table_stride_b, LL_F16, req_width, seq_width, counters, nullptr, stream, true, k_scale,
                  v_scale);
}

// Decode-step attention in ONE launch: rope(q, k_new) + KV scatter of the new token + flash_decoding
// (= ll_rope_kv_update with position-indexed tables, then ll_flash_decoding; same values).  q is
// read un-rotated and NOT written back; kv_new rows are [2*hkv, d] (K heads first); the pool views
// k_cache / v_cache are written at row select_index[b].  Requires: counters (see ll_flash_decoding),
// d >= 64, hq/hkv <= 16, cos/sin of the q dtype, distinct select rows, and
// table[b_req_idx[b], b_seq_len[b]-1] == select_index[b].
// q_norm_weight / k_norm_weight ([d], q dtype; both or neither, d == 128): Qwen3's per-head RMSNorm of q and of the
// new K heads (eps = norm_eps) runs in front of the rotation -- the values of ll_skip_rmsnorm on the [.., d] views.
extern "C" int ll_decode_attention(void* out, const void* q, const void* kv_new, int64_t kv_row_stride,
                                   const void* cos_t, const void* sin_t, int64_t cs_row_stride,
                                   const int64_t* positions, const void* select_index, int sel_width,
                                   void* k_cache, void* v_cache, const int32_t* table, const void* b_req_idx,
                                   const void* b_seq_len, float* mid_o, float* mid_lse, int batch, int hq, int hkv,
                                   int d, int64_t max_len, float qk_scale, int64_t q_stride_b, int64_t q_stride_h,
                                   int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                   int64_t o_stride_b, int64_t o_stride_h, int64_t table_stride_b, int dtype,
                                   int req_width, int seq_width, int32_t* counters, const void* q_norm_weight,
                                   const void* k_norm_weight, float norm_eps, void* stream) {
  if (!kv_new || !cos_t || !sin_t || !positions || !select_index || !counters) return LL_ERR_ARG;
  if ((q_norm_weight != nullptr) != (k_norm_weight != nullptr) || !ll_aligned16(q_norm_weight) || !ll_aligned16(k_norm_weight))
    return LL_ERR_ARG;
  if (sel_width != LL_I32 && sel_width != LL_I64) return LL_ERR_DTYPE;
  if ((kv_row_stride | cs_row_stride) % 8 != 0 || !ll_aligned16(kv_new) || !ll_aligned16(cos_t) || !ll_aligned16(sin_t))
    return LL_ERR_ARG;
  FdRope rp{(const uint16_t*)kv_new, kv_row_stride, (const uint16_t*)cos_t, (const uint16_t*)sin_t, cs_row_stride,
            positions, select_index, sel_width, (uint16_t*)k_cache, (uint16_t*)v_cache, nullptr, 0, 0, 0, nullptr};
  rp.qnw = (const uint16_t*)q_norm_weight;
  rp.knw = (const uint16_t*)k_norm_weight;
  rp.nrm_eps = norm_eps;
  return fd_entry(out, q, k_cache, v_cache, table, b_req_idx, b_seq_len, mid_o, mid_lse, batch, hq, hkv, d, max_len,
                  qk_scale, q_stride_b, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_b,
                  o_stride_h, table_stride_b, dtype, req_width, seq_width, counters, &rp, stream);
}

// ll_decode_attention whose q / k_new / v_new come as the fp32 split-K partials of the fused q|k|v projection
// (ll_w4a16_matmul_prepacked epilogue 2): qkv_partials [s_count][batch][row_width] with row_width = (hq + 2 hkv) * d
// (q heads, K heads, V heads), qkv_bias [row_width] (projection bias, dtype of the pool) or NULL.  x = fp16(sum of the
// partials + bias) -- the value the projection would have stored -- then exactly ll_decode_attention.  Served for
// contexts of 2..8 partitions (129..1024 tokens); LL_ERR_SHAPE otherwise: finish the sums and call ll_decode_attention.
// int32_a_scale [batch] / int32_w_scale [row_width] (both or neither; fp16, d == 128): the planes are the EXACT int32 sums of
// a smoothquant projection (ll_dense_partials wfmt 3) and x = fp16(((float)sum * a_scale[row]) * w_scale[col] (+ bias)) --
// what ll_w8a8_matmul stores.
extern "C" int ll_decode_attention_partials(void* out, const float* qkv_partials, int s_count, const void* qkv_bias,
                                            const void* cos_t, const void* sin_t, int64_t cs_row_stride,
                                            const int64_t* positions, const void* select_index, int sel_width,
                                            void* k_cache, void* v_cache, const int32_t* table, const void* b_req_idx,
                                            const void* b_seq_len, int batch, int hq, int hkv, int d, int64_t max_len,
                                            float qk_scale, int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t,
                                            int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                            int64_t table_stride_b, int dtype, int req_width, int seq_width,
                                            const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
                                            const float* int32_a_scale, const float* int32_w_scale, void* stream) {
  if (!qkv_partials || !cos_t || !sin_t || !positions || !select_index) return LL_ERR_ARG;
  if ((int32_a_scale != nullptr) != (int32_w_scale != nullptr) || !ll_aligned16(int32_w_scale)) return LL_ERR_ARG;
  if ((q_norm_weight != nullptr) != (k_norm_weight != nullptr) || !ll_aligned16(q_norm_weight) || !ll_aligned16(k_norm_weight))
    return LL_ERR_ARG;
  if (sel_width != LL_I32 && sel_width != LL_I64) return LL_ERR_DTYPE;
  if (cs_row_stride % 8 != 0 || !ll_aligned16(qkv_partials) || !ll_aligned16(cos_t) || !ll_aligned16(sin_t) || d % 4 != 0)
    return LL_ERR_ARG;
  const int64_t row_w = (int64_t)(hq + 2 * hkv) * d;
  static int32_t dummy_counters = 0;  // the grouped form never touches the counters; fd_entry only wants a non-null pointer
  FdRope rp{nullptr, 0, (const uint16_t*)cos_t, (const uint16_t*)sin_t, cs_row_stride, positions, select_index,
            sel_width, (uint16_t*)k_cache, (uint16_t*)v_cache, qkv_partials, s_count, (int64_t)batch * row_w, row_w,
            (const uint16_t*)qkv_bias};
  rp.qnw = (const uint16_t*)q_norm_weight;
  rp.knw = (const uint16_t*)k_norm_weight;
  rp.nrm_eps = norm_eps;
  rp.q_as = int32_a_scale;
  rp.q_ws = int32_w_scale;
  // q pointer / strides are unused in this mode; pass the pool (aligned, non-null) to satisfy the argument checks
  return fd_entry(out, k_cache, k_cache, v_cache, table, b_req_idx, b_seq_len, nullptr, nullptr, batch, hq, hkv, d, max_len,
                  qk_scale, 8, 8, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_b, o_stride_h, table_stride_b,
                  dtype, req_width, seq_width, &dummy_counters, &rp, stream);
}
