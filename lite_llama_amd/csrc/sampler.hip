// Sampler row: reference lite_llama/engine/sampler.py:77-137,199-270.
//   ll_repetition_penalty -- apply_repetition_penalty (:77-115): one scatter over the generated span.
//   ll_sample_top_p       -- softmax(logits / T) + nucleus filter (:118-137) + one draw per row from a
//                            caller-supplied uniform number, WITHOUT the reference's full sort.
// The nucleus is the set of tokens whose strictly-more-probable mass does not exceed top_p: with
// e_i = exp(z_i - max) and f(t) = sum of e_i > t, a token is kept iff f(e_i) <= top_p * sum(e).  The
// smallest t with f(t) <= P is itself a data value (f only jumps at data values); positive floats
// order like their bit patterns, so a bisection over the pattern (<= 30 passes of a cheap
// compare-accumulate over the row) finds it exactly; tokens
// tied at that value are kept in ascending index order while the mass before them fits (the
// reference leaves tie order to torch.sort); the draw is the inverse CDF over the kept tokens in index order (same distribution as the reference's
// torch.multinomial over the sorted row, different random stream).  One 1024-thread workgroup per row.
#include "common.h"

template <int DT>
__device__ __forceinline__ float ld_logit(const void* p, int64_t i) {
  if constexpr (DT == LL_F32) return ((const float*)p)[i];
  else return to_f32<DT>(((const uint16_t*)p)[i]);
}
template <int DT>
__device__ __forceinline__ void st_logit(void* p, int64_t i, float v) {
  if constexpr (DT == LL_F32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = from_f32<DT>(v);
}

// ------------------------------------------------------------------------------------------- //
// repetition penalty: out = logits, then for every (b, j) with mask: out[b, tok] = penalised value of
// the ORIGINAL logit (so repeated tokens are penalised once -- sampler.py:99-101).  out != logits.
// ------------------------------------------------------------------------------------------- //
// grid = (chunks, batch): a workgroup copies its chunk of the row and then applies the penalties of
// the span tokens that fall INTO that chunk (so no other workgroup's copy can overwrite them).
template <int DTI, int DTO>
__global__ __launch_bounds__(256) void rep_penalty_kernel(void* __restrict__ out, const void* __restrict__ logits,
                                                          const int64_t* __restrict__ ids,
                                                          const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ pen_rows, float pen_scalar,
                                                          int64_t vocab, int64_t span, int64_t lstride,
                                                          int64_t ostride, int64_t istride, int64_t mstride) {
  const int64_t b = blockIdx.y;
  const int64_t per = (vocab + gridDim.x - 1) / gridDim.x;
  const int64_t c_lo = (int64_t)blockIdx.x * per;
  const int64_t c_hi = c_lo + per < vocab ? c_lo + per : vocab;
  for (int64_t i = c_lo + threadIdx.x; i < c_hi; i += 256)
    st_logit<DTO>(out, b * ostride + i, ld_logit<DTI>(logits, b * lstride + i));
  __syncthreads();  // this chunk's copy is complete before any penalised value lands on it
  const float pen = pen_rows ? pen_rows[b] : pen_scalar;
  for (int64_t j = threadIdx.x; j < span; j += 256) {
    if (!mask[b * mstride + j]) continue;
    const int64_t tok = ids[b * istride + j];
    if (tok < c_lo || tok >= c_hi) continue;
    const float x = ld_logit<DTI>(logits, b * lstride + tok);
    st_logit<DTO>(out, b * ostride + tok, x < 0.f ? x * pen : x / pen);
  }
}

extern "C" int ll_repetition_penalty(void* out, const void* logits, const int64_t* token_ids, const void* mask,
                                     const float* penalty_rows, float penalty_scalar, int64_t batch,
                                     int64_t vocab, int64_t span, int64_t logits_stride, int64_t out_stride,
                                     int64_t ids_stride, int64_t mask_stride, int in_dtype, int out_dtype,
                                     void* stream) {
  if (batch < 0 || vocab <= 0 || span < 0) return LL_ERR_SHAPE;
  if (!out || !logits || out == logits || (span > 0 && (!token_ids || !mask))) return LL_ERR_ARG;
  if (batch == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned chunks = (unsigned)((vocab + 8191) / 8192);  // ~8 K logits per workgroup
  if (chunks > 64) chunks = 64;
#define LL_RP(DTI, DTO)                                                                                   \
  rep_penalty_kernel<DTI, DTO><<<dim3(chunks, (unsigned)batch), 256, 0, st>>>(                            \
      out, logits, token_ids, (const uint8_t*)mask, penalty_rows, penalty_scalar, vocab, span, logits_stride, \
      out_stride, ids_stride, mask_stride)
  if (in_dtype == LL_F32 && out_dtype == LL_F32) LL_RP(LL_F32, LL_F32);
  else if (in_dtype == LL_F16 && out_dtype == LL_F16) LL_RP(LL_F16, LL_F16);
  else if (in_dtype == LL_F16 && out_dtype == LL_F32) LL_RP(LL_F16, LL_F32);
  else if (in_dtype == LL_BF16 && out_dtype == LL_BF16) LL_RP(LL_BF16, LL_BF16);
  else if (in_dtype == LL_BF16 && out_dtype == LL_F32) LL_RP(LL_BF16, LL_F32);
  else return LL_ERR_DTYPE;
#undef LL_RP
  return LL_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------- //
// nucleus sampling
// ------------------------------------------------------------------------------------------- //
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sh[i];  // same order in every thread: identical result everywhere
  return t;
}
__device__ __forceinline__ float block_max_1024(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = sh[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) t = fmaxf(t, sh[i]);
  return t;
}

// e_i = exp(z_i - max) through the hardware exp2 (one v_exp_f32, ~1 ulp): the whole kernel is a few
// dozen passes over the row and is instruction-issue bound, so the libm expf (~20 instructions) would
// dominate it; equal logits still give equal e, which is all the tie rule needs.
__device__ __forceinline__ float fast_e(float l, float k, float mk) { return __builtin_amdgcn_exp2f(l * k - mk); }

// Row layout: wave w owns the contiguous segment [w*seg, (w+1)*seg) and walks it 64 tokens at a time
// with lane = token offset, so every load is one coalesced 128-B (fp16) request; token order is
// (wave, iteration, lane), which is what the final inverse-CDF walk needs.
template <int DT>
__global__ __launch_bounds__(1024) void top_p_sample_kernel(int64_t* __restrict__ out, const void* __restrict__ logits,
                                                            const float* __restrict__ temperature,
                                                            const float* __restrict__ top_p,
                                                            const float* __restrict__ uniform,
                                                            const uint8_t* __restrict__ greedy, int64_t vocab,
                                                            int64_t stride) {
  __shared__ float sh[16];
  __shared__ float gw[16], cw[16];
  __shared__ int64_t s_idx[16];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t base = b * stride;
  const int64_t seg = ((vocab + 15) / 16 + 63) / 64 * 64;
  const int64_t s_lo = w * seg < vocab ? w * seg : vocab;
  const int64_t s_hi = s_lo + seg < vocab ? s_lo + seg : vocab;

  if (greedy && greedy[b]) {  // temperature == 0 rows: first maximum, like torch.argmax
    float best = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = s_lo + lane; i < s_hi; i += 64) {
      const float v = ld_logit<DT>(logits, base + i);
      if (bi == INT64_MAX || v > best) {  // per-lane indices ascend: strict > keeps the first max
        best = v;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int64_t oi = __shfl_xor(bi, off, 64);
      if (oi != INT64_MAX && (bi == INT64_MAX || ov > best || (ov == best && oi < bi))) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      sh[w] = best;
      s_idx[w] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int j = 1; j < 16; ++j)
        if (s_idx[j] != INT64_MAX && (bi == INT64_MAX || sh[j] > best || (sh[j] == best && s_idx[j] < bi))) {
          best = sh[j];
          bi = s_idx[j];
        }
      out[b] = bi;
    }
    return;
  }

  // z = logit / T evaluated as logit * (1 / T): equal logits stay equal
  const float k = 1.4426950408889634f / temperature[b];  // log2(e) / T
  float mk = -INFINITY;
  for (int64_t i = s_lo + lane; i < s_hi; i += 64) mk = fmaxf(mk, ld_logit<DT>(logits, base + i) * k);
  mk = block_max_1024(mk, sh);
  float s = 0.f;
  for (int64_t i = s_lo + lane; i < s_hi; i += 64) s += fast_e(ld_logit<DT>(logits, base + i), k, mk);
  const float S = block_sum_1024(s, sh);
  const float P = top_p[b] * S;

  // tau = the smallest kept value = smallest t with (mass of e > t) <= P.  Positive floats order
  // like their bit patterns: bisection over the pattern in [0, 1.0f] (f(1.0) = 0 <= P always holds).
  uint32_t blo = 0, bhi = 0x3F800000u;
  float G = 0.f;  // mass strictly above the current upper bound
  while (blo < bhi) {
    const uint32_t mid = blo + ((bhi - blo) >> 1);
    const float t = __uint_as_float(mid);
    float a = 0.f;
#pragma unroll 8
    for (int64_t i = s_lo + lane; i < s_hi; i += 64) {
      const float e = fast_e(ld_logit<DT>(logits, base + i), k, mk);
      a += e > t ? e : 0.f;
    }
    a = block_sum_1024(a, sh);
    if (a <= P) {
      bhi = mid;
      G = a;
    } else {
      blo = mid + 1;
    }
  }
  const float tau = __uint_as_float(bhi);

  // per-segment kept mass; tokens tied at tau are kept in ascending index order while the mass
  // before them fits (the first n_tie of the row)
  float g_mine = 0.f, c_mine = 0.f;
  for (int64_t i = s_lo + lane; i < s_hi; i += 64) {
    const float e = fast_e(ld_logit<DT>(logits, base + i), k, mk);
    g_mine += e > tau ? e : 0.f;
    c_mine += e == tau ? 1.f : 0.f;
  }
  g_mine = wave_sum(g_mine);
  c_mine = wave_sum(c_mine);  // counts <= vocab: exact in fp32
  __syncthreads();
  if (lane == 0) {
    gw[w] = g_mine;
    cw[w] = c_mine;
  }
  __syncthreads();
  float C = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) C += cw[j];
  float n_tie = C;  // tau == 0 (top_p = 1 keeps everything): nothing to ration
  if (tau > 0.f && C > 0.f) {
    n_tie = floorf((P - G) / tau) + 1.f;
    n_tie = n_tie < 1.f ? 1.f : (n_tie > C ? C : n_tie);
  }
  // every thread walks the 16 segment totals in the same order: Z, the chosen segment, the mass and
  // the kept ties before it
  float Z = 0.f, ties_seen = 0.f;
  float seg_mass[16], seg_ties[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float kt = n_tie - ties_seen;
    kt = kt < 0.f ? 0.f : (kt > cw[j] ? cw[j] : kt);
    seg_ties[j] = kt;
    seg_mass[j] = gw[j] + tau * kt;
    Z += seg_mass[j];
    ties_seen += cw[j];
  }
  const float target = uniform[b] * Z;
  int pick_w = -1, last_w = -1;
  float before = 0.f, run = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (seg_mass[j] > 0.f) {
      last_w = j;
      if (pick_w < 0 && target < run + seg_mass[j]) {
        pick_w = j;
        before = run;
      }
    }
    run += seg_mass[j];
  }
  bool overflow = false;
  if (pick_w < 0) {  // target landed on / beyond the total because of rounding (u ~ 1)
    pick_w = last_w;
    overflow = true;
  }
  if (w != pick_w) return;

  // the chosen wave walks its segment in index order
  float ties_left = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) ties_left = j == pick_w ? seg_ties[j] : ties_left;
  float acc_run = before;
  int64_t pick = -1, last_kept = -1;
  for (int64_t i0 = s_lo; i0 < s_hi && pick < 0; i0 += 64) {
    const int64_t i = i0 + lane;
    const float e = i < s_hi ? fast_e(ld_logit<DT>(logits, base + i), k, mk) : -1.f;
    const bool is_tie = e == tau;
    const unsigned long long tb = __ballot(is_tie);
    const float rank = (float)__popcll(tb & ((1ull << lane) - 1ull));
    const bool keep = e > tau || (is_tie && rank < ties_left);
    float incl = keep ? e : 0.f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const unsigned long long kb = __ballot(keep);
    if (kb) last_kept = i0 + (63 - __builtin_clzll(kb));
    const unsigned long long hit = __ballot(keep && acc_run + incl > target);
    if (hit && !overflow) {
      pick = i0 + (__builtin_ffsll((long long)hit) - 1);
      break;
    }
    acc_run += __shfl(incl, 63, 64);
    float kept_ties = (float)__popcll(tb);
    kept_ties = kept_ties < ties_left ? kept_ties : ties_left;
    ties_left -= kept_ties;
  }
  if (lane == 0) out[b] = pick >= 0 ? pick : last_kept;  // fp rounding: the segment's last kept token
}

extern "C" int ll_sample_top_p(int64_t* out, const void* logits, const float* temperature, const float* top_p,
                               const float* uniform, const void* greedy, int64_t batch, int64_t vocab,
                               int64_t stride, int dtype, void* stream) {
  if (batch < 0 || vocab <= 0) return LL_ERR_SHAPE;
  if (!out || !logits || !temperature || !top_p || !uniform) return LL_ERR_ARG;
  if (batch == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)batch);
  if (dtype == LL_F32)
    top_p_sample_kernel<LL_F32><<<grid, 1024, 0, st>>>(out, logits, temperature, top_p, uniform, (const uint8_t*)greedy, vocab, stride);
  else if (dtype == LL_F16)
    top_p_sample_kernel<LL_F16><<<grid, 1024, 0, st>>>(out, logits, temperature, top_p, uniform, (const uint8_t*)greedy, vocab, stride);
  else if (dtype == LL_BF16)
    top_p_sample_kernel<LL_BF16><<<grid, 1024, 0, st>>>(out, logits, temperature, top_p, uniform, (const uint8_t*)greedy, vocab, stride);
  else return LL_ERR_DTYPE;
  return LL_LAUNCH_CHECK();
}
