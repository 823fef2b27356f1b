// flash_attention2_no_pad (a6) -- reference lite_llama/kernels/flashattention2_nopad.py:45-231.
// Varlen causal prefill over the freshly projected q/k/v (not the cache): exp2 softmax
// (sm_scale carries log2 e), masked score -1.0e8, P rounded to the storage dtype before PV,
// fp32 accumulators, rows past b_seq_len are not written.
//
// Same MFMA formulation as the decode kernel: one wave owns 16 consecutive query rows of one
// (sequence, head); S^T = K.Q^T puts keys in MFMA rows and queries in columns, so the online
// softmax is lane-local per query, and the S^T accumulators are directly the P^T B-operand of
// O^T = V^T.P^T.  K fragments come straight from global memory (L2-shared between the waves
// of a head), V is staged per 32-key tile through LDS to be read key-major.
#include <stdlib.h>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

__device__ __forceinline__ int64_t pa_load_idx(const void* p, int64_t i, int w) {
  return w == LL_I32 ? (int64_t)((const int32_t*)p)[i] : ((const int64_t*)p)[i];
}

// grid = (ceil(max_seq_len / 16), hq, batch), block = 64 (one wave)
template <int DT, int D>
__global__ __launch_bounds__(64) void fa_prefill(
    uint16_t* __restrict__ out, const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
    const uint16_t* __restrict__ v, const void* __restrict__ b_start_loc, const void* __restrict__ b_seq_len,
    int hq, int hkv, float sm_scale, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st,
    int64_t v_sh, int64_t o_st, int64_t o_sh, int start_w, int seq_w) {
  constexpr int NS = D / 32, DQ = D / 4, NT = D / 16, VSTR = D + 8;
  __shared__ __attribute__((aligned(16))) uint16_t lds_v[32 * VSTR];

  const int lane = threadIdx.x;
  const int t = lane & 15, c = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int kvh = head / (hq / hkv);
  const int64_t seq_len = pa_load_idx(b_seq_len, b, seq_w);
  const int64_t start = pa_load_idx(b_start_loc, b, start_w);
  const int64_t q0 = (int64_t)blockIdx.x * 16;
  if (q0 >= seq_len) return;

  // Q^T fragments: lane (query t, group c); rows past the sequence end read row seq_len-1
  // (never stored)
  const int64_t qrow = (q0 + t < seq_len) ? q0 + t : seq_len - 1;
  Q4 qf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
    qf[s] = *reinterpret_cast<const Q4*>(q + (start + qrow) * q_st + (int64_t)head * q_sh + c * DQ + s * 8);

  float m_i = -INFINITY, d_i = 0.f;
  f32x4 ot[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) ot[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t kend = (q0 + 16 < seq_len) ? q0 + 16 : seq_len;  // causal: keys < min(q0 + 16, seq_len)
  for (int64_t p0 = 0; p0 < kend; p0 += 32) {
    const int64_t ka_i = p0 + t, kb_i = p0 + 16 + t;
    const bool okA = ka_i < kend, okB = kb_i < kend;
    const int64_t ra = start + (okA ? ka_i : 0), rb = start + (okB ? kb_i : 0);
    const uint16_t* kA = k + ra * k_st + (int64_t)kvh * k_sh + c * DQ;
    const uint16_t* kB = k + rb * k_st + (int64_t)kvh * k_sh + c * DQ;
    const uint16_t* vA = v + ra * v_st + (int64_t)kvh * v_sh + c * DQ;
    const uint16_t* vB = v + rb * v_st + (int64_t)kvh * v_sh + c * DQ;
    Q4 ka[NS], kb[NS], va[NS], vb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      ka[s] = *reinterpret_cast<const Q4*>(kA + s * 8);
      kb[s] = *reinterpret_cast<const Q4*>(kB + s * 8);
      va[s] = *reinterpret_cast<const Q4*>(vA + s * 8);
      vb[s] = *reinterpret_cast<const Q4*>(vB + s * 8);
    }
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) sa = mfma16<DT>(ka[s], qf[s], sa);
#pragma unroll
    for (int s = 0; s < NS; ++s) sb = mfma16<DT>(kb[s], qf[s], sb);

    // rows = keys p0 + 4c + r (+16), col = query q0 + t
    const int64_t qi = q0 + t;
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t k0i = p0 + 4 * c + r, k1i = p0 + 16 + 4 * c + r;
      // reference: where(causal, qk * sm_scale, -1.0e8); keys past the block end are causal-masked too
      sc[r] = (k0i <= qi && k0i < kend) ? sa[r] * sm_scale : -1.0e8f;
      sc[4 + r] = (k1i <= qi && k1i < kend) ? sb[r] * sm_scale : -1.0e8f;
      mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_i, mx);
    const float alpha = exp2f(m_i - m_new);
    float p[8];
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = exp2f(sc[j] - m_new);
      ps += p[j];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    d_i = d_i * alpha + ps;
    m_i = m_new;
    Q4 pf;
    pf.x = pack2<DT>(p[0], p[1]);
    pf.y = pack2<DT>(p[2], p[3]);
    pf.z = pack2<DT>(p[4], p[5]);
    pf.w = pack2<DT>(p[6], p[7]);

    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      *reinterpret_cast<Q4*>(&lds_v[t * VSTR + c * DQ + s * 8]) = okA ? va[s] : Q4{0, 0, 0, 0};
      *reinterpret_cast<Q4*>(&lds_v[(16 + t) * VSTR + c * DQ + s * 8]) = okB ? vb[s] : Q4{0, 0, 0, 0};
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) {
      uint32_t w[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j0 = 2 * jj, j1 = 2 * jj + 1;
        const int r0 = 16 * (j0 >> 2) + 4 * c + (j0 & 3);
        const int r1 = 16 * (j1 >> 2) + 4 * c + (j1 & 3);
        const uint32_t lo = lds_v[r0 * VSTR + dt * 16 + t];
        const uint32_t hi = lds_v[r1 * VSTR + dt * 16 + t];
        w[jj] = lo | (hi << 16);
      }
      const Q4 vf = {w[0], w[1], w[2], w[3]};
      f32x4 acc = ot[dt];
      acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
      ot[dt] = mfma16<DT>(vf, pf, acc);
    }
  }

  if (q0 + t < seq_len) {
    const float inv = 1.0f / d_i;
    uint16_t* o = out + (start + q0 + t) * o_st + (int64_t)head * o_sh;
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) {
      uint2 pk;
      pk.x = pack2<DT>(ot[dt][0] * inv, ot[dt][1] * inv);
      pk.y = pack2<DT>(ot[dt][2] * inv, ot[dt][3] * inv);
      *reinterpret_cast<uint2*>(o + dt * 16 + 4 * c) = pk;
    }
  }
}

// Second form (D = 64 / 128): a four-wave workgroup owns 64 * QB consecutive queries of one (sequence, head) and walks
// the keys in tiles of 64.  A tile's K and V rows are fetched ONCE per workgroup (global -> registers -> LDS, the next
// tile's loads in flight under the current tile's MFMAs) and shared by the four waves: 64 * QB queries per byte of K/V
// instead of 16 -- the one-wave form re-reads K/V per 16 queries and saturates the CU's memory pipeline (4.8 TB/s of
// L2 traffic at batch 64 x 512).  Each wave keeps 16 * QB query columns: K fragments are read from LDS once per tile
// and used for QB query blocks, V^T fragments come from transposing LDS reads (ds_read_b64_tr_b16), the online softmax
// runs once per 64 keys (one rescale of O per tile, skipped when no running maximum moved).
template <int DT, int D, int QB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void fa_prefill2(
    uint16_t* __restrict__ out, const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
    const uint16_t* __restrict__ v, const void* __restrict__ b_start_loc, const void* __restrict__ b_seq_len,
    int hq, int hkv, float sm_scale, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st,
    int64_t v_sh, int64_t o_st, int64_t o_sh, int start_w, int seq_w) {
  constexpr int NS = D / 32, NT = D / 16, STR = D + 8, TK = 64;
  constexpr int TPR = D / 8, RPP = 256 / TPR, PASSES = TK / RPP;  // 16-byte pieces per row, rows per pass
  __shared__ __attribute__((aligned(16))) uint16_t lds_k[TK * STR];
  __shared__ __attribute__((aligned(16))) uint16_t lds_v[TK * STR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, c = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int kvh = head / (hq / hkv);
  const int64_t seq_len = pa_load_idx(b_seq_len, b, seq_w);
  const int64_t start = pa_load_idx(b_start_loc, b, start_w);
  const int64_t q0 = (int64_t)(gridDim.x - 1 - blockIdx.x) * (64 * QB);  // the longest key ranges are dispatched first
  if (q0 >= seq_len) return;
  const int64_t qw = q0 + wave * (16 * QB);  // this wave's first query

  Q4 qf[QB][NS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t qi = qw + qb * 16 + t;
    const int64_t qrow = qi < seq_len ? qi : seq_len - 1;  // rows past the end: a valid row, never stored
#pragma unroll
    for (int s = 0; s < NS; ++s)
      qf[qb][s] = *reinterpret_cast<const Q4*>(q + (start + qrow) * q_st + (int64_t)head * q_sh + s * 32 + c * 8);
  }
  float m_i[QB], d_i[QB];
  f32x4 ot[QB][NT];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_i[qb] = -INFINITY;
    d_i[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) ot[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int64_t kend = q0 + 64 * QB < seq_len ? q0 + 64 * QB : seq_len;  // causal: the workgroup needs keys < kend
  const int lrow = tid / TPR, lcol = (tid % TPR) * 8;
  i32x4 kr[PASSES], vr[PASSES];  // (vector type, not the Q4 struct: the struct copy into LDS kept the array in scratch)
  auto fetch = [&](int64_t p0) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      int64_t key = p0 + ps * RPP + lrow;
      if (key >= seq_len) key = seq_len - 1;
      kr[ps] = *reinterpret_cast<const i32x4*>(k + (start + key) * k_st + (int64_t)kvh * k_sh + lcol);
      vr[ps] = *reinterpret_cast<const i32x4*>(v + (start + key) * v_st + (int64_t)kvh * v_sh + lcol);
    }
  };
  auto stage = [&](int64_t p0) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = ps * RPP + lrow;
      *reinterpret_cast<i32x4*>(&lds_k[row * STR + lcol]) = kr[ps];
      // rows past the sequence end are multiplied by p = 0: they must not hold NaN / Inf bit patterns
      *reinterpret_cast<i32x4*>(&lds_v[row * STR + lcol]) = p0 + row < seq_len ? vr[ps] : i32x4{0, 0, 0, 0};
    }
  };
  const uint16_t* vtr = lds_v + (4 * c + (t >> 2)) * STR + (t & 3) * 4;

  fetch(0);
  for (int64_t p0 = 0; p0 < kend; p0 += TK) {
    __syncthreads();  // the previous tile's fragment reads are done
    stage(p0);
    __syncthreads();
    fetch(p0 + TK < kend ? p0 + TK : p0);  // in flight under this tile's arithmetic (unconditional: a branch around
                                           // loads parks the prefetch registers in scratch)
    if (p0 >= qw + 16 * QB) continue;    // every key of the tile lies after this wave's last query
    // S^T blocks for all QB query blocks from ONE read of each K fragment: rows = keys p0 + 16 kb + 4c + r, column =
    // query qw + 16 qb + t.  (A query block that lies entirely before the tile gets all-masked scores: p = 0, alpha = 1.)
    f32x4 sacc[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const Q4 kf = *reinterpret_cast<const Q4*>(&lds_k[(kb * 16 + t) * STR + s * 32 + c * 8]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mfma16<DT>(kf, qf[qb][s], sacc[qb][kb]);
      }
    }
    Q4 pf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int64_t qfirst = qw + qb * 16;
      // scores stay raw until the exponent: p = exp2(s * scale - m) is one fma + one v_exp per key (scale > 0, so the
      // maximum is taken over the raw scores); masked keys hold the raw value whose scaled form is the reference's -1e8
      float sc[16];
      float mx = -INFINITY;
      if (p0 + TK - 1 <= qfirst) {  // whole tile at or before the block's first query: no mask (wave-uniform)
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = sacc[qb][j >> 2][j & 3];
      } else {
        const int64_t qi = qfirst + t;
        const float masked = -1.0e8f / sm_scale;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int64_t key = p0 + 16 * (j >> 2) + 4 * c + (j & 3);
          sc[j] = key <= qi ? sacc[qb][j >> 2][j & 3] : masked;  // reference: where(causal, qk * scale, -1e8)
        }
      }
#pragma unroll
      for (int j = 0; j < 16; j += 2) mx = fmaxf(mx, fmaxf(sc[j], sc[j + 1]));
      mx *= sm_scale;
      mx = rows_max(mx);
      const float m_new = fmaxf(m_i[qb], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_i[qb] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        sc[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], sm_scale, -m_new));
        psum += sc[j];
      }
      psum = rows_sum(psum);
      d_i[qb] = d_i[qb] * alpha + psum;
      m_i[qb] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // some query's maximum moved: rescale O
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
          ot[qb][dt][0] *= alpha; ot[qb][dt][1] *= alpha; ot[qb][dt][2] *= alpha; ot[qb][dt][3] *= alpha;
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // 32 keys per PV step: P^T k-slot j <-> key 32 hf + 16 (j >> 2) + 4c + (j & 3)
        pf[qb][hf].x = pack2<DT>(sc[8 * hf + 0], sc[8 * hf + 1]);
        pf[qb][hf].y = pack2<DT>(sc[8 * hf + 2], sc[8 * hf + 3]);
        pf[qb][hf].z = pack2<DT>(sc[8 * hf + 4], sc[8 * hf + 5]);
        pf[qb][hf].w = pack2<DT>(sc[8 * hf + 6], sc[8 * hf + 7]);
      }
    }
    // O^T += V^T P^T: each V^T fragment (transposing LDS read) feeds all QB query blocks
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        const s16x4 tlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vtr + (32 * hf) * STR + dt * 16));
        const s16x4 thi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vtr + (32 * hf + 16) * STR + dt * 16));
        Q4 vf;
        vf.x = (uint32_t)(uint16_t)tlo.x | ((uint32_t)(uint16_t)tlo.y << 16);
        vf.y = (uint32_t)(uint16_t)tlo.z | ((uint32_t)(uint16_t)tlo.w << 16);
        vf.z = (uint32_t)(uint16_t)thi.x | ((uint32_t)(uint16_t)thi.y << 16);
        vf.w = (uint32_t)(uint16_t)thi.z | ((uint32_t)(uint16_t)thi.w << 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) ot[qb][dt] = mfma16<DT>(vf, pf[qb][hf], ot[qb][dt]);
      }
    }
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int64_t qi = qw + qb * 16 + t;
    if (qi < seq_len) {
      const float inv = 1.0f / d_i[qb];
      uint16_t* o = out + (start + qi) * o_st + (int64_t)head * o_sh;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        uint2 pk;
        pk.x = pack2<DT>(ot[qb][dt][0] * inv, ot[qb][dt][1] * inv);
        pk.y = pack2<DT>(ot[qb][dt][2] * inv, ot[qb][dt][3] * inv);
        *reinterpret_cast<uint2*>(o + dt * 16 + 4 * c) = pk;
      }
    }
  }
}

template <int DT>
static int launch_fa(void* out, const void* q, const void* k, const void* v, const void* start, const void* seq,
                     int batch, int hq, int hkv, int d, int64_t max_seq_len, float sm_scale, int64_t q_st,
                     int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh, int64_t o_st,
                     int64_t o_sh, int start_w, int seq_w, hipStream_t st) {
  if (d == 64 || d == 128) {
    static const int qb_env = getenv("LL_FA_QB") ? atoi(getenv("LL_FA_QB")) : 0;  // debug knob, read once
    const int qb = qb_env == 1 || qb_env == 2 ? qb_env : (max_seq_len > 64 ? 2 : 1);
    dim3 grid2((unsigned)((max_seq_len + 64 * qb - 1) / (64 * qb)), (unsigned)hq, (unsigned)batch);
#define LL_FA2(DD, QQ)                                                                                          \
  fa_prefill2<DT, DD, QQ><<<grid2, 256, 0, st>>>((uint16_t*)out, (const uint16_t*)q, (const uint16_t*)k,        \
                                                 (const uint16_t*)v, start, seq, hq, hkv, sm_scale, q_st, q_sh, \
                                                 k_st, k_sh, v_st, v_sh, o_st, o_sh, start_w, seq_w)
    if (d == 64) {
      if (qb == 2) LL_FA2(64, 2); else LL_FA2(64, 1);
    } else {
      if (qb == 2) LL_FA2(128, 2); else LL_FA2(128, 1);
    }
#undef LL_FA2
    return LL_LAUNCH_CHECK();
  }
  dim3 grid((unsigned)((max_seq_len + 15) / 16), (unsigned)hq, (unsigned)batch);
#define LL_FA(DD)                                                                                        \
  fa_prefill<DT, DD><<<grid, 64, 0, st>>>((uint16_t*)out, (const uint16_t*)q, (const uint16_t*)k,        \
                                          (const uint16_t*)v, start, seq, hq, hkv, sm_scale, q_st, q_sh, \
                                          k_st, k_sh, v_st, v_sh, o_st, o_sh, start_w, seq_w)
  switch (d) {
    case 32: LL_FA(32); break;
    case 64: LL_FA(64); break;
    case 128: LL_FA(128); break;
    case 256: LL_FA(256); break;
    default: return LL_ERR_SHAPE;
  }
#undef LL_FA
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_flash_attention_nopad(void* out, const void* q, const void* k, const void* v,
                                        const void* b_start_loc, const void* b_seq_len, int batch, int hq,
                                        int hkv, int d, int64_t max_seq_len, float sm_scale, int64_t q_stride_t,
                                        int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                                        int64_t v_stride_t, int64_t v_stride_h, int64_t o_stride_t,
                                        int64_t o_stride_h, int dtype, int start_width, int seq_width,
                                        void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if ((start_width | seq_width) & ~1) return LL_ERR_DTYPE;
  if (batch < 0 || hq <= 0 || hkv <= 0 || hq % hkv != 0 || max_seq_len < 0) return LL_ERR_SHAPE;
  if (batch == 0 || max_seq_len == 0) return LL_OK;
  if ((q_stride_t | q_stride_h | k_stride_t | k_stride_h | v_stride_t | v_stride_h | o_stride_t | o_stride_h) % 8 != 0 ||
      !ll_aligned16(q) || !ll_aligned16(k) || !ll_aligned16(v) || !ll_aligned16(out))
    return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16)
    return launch_fa<LL_F16>(out, q, k, v, b_start_loc, b_seq_len, batch, hq, hkv, d, max_seq_len, sm_scale,
                             q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t,
                             o_stride_h, start_width, seq_width, st);
  return launch_fa<LL_BF16>(out, q, k, v, b_start_loc, b_seq_len, batch, hq, hkv, d, max_seq_len, sm_scale,
                            q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t,
                            o_stride_h, start_width, seq_width, st);
}
