// KV-cache bookkeeping kernels (token-attention layout, block_size = 1).
//   update_kv_buffer -- reference lite_llama/kernels/update_kv_buffer.py:15-89
//   update_kv_index  -- reference lite_llama/kernels/update_kv_index.py:15-88
// Pure integer / byte moves: bit-exact by construction.
#include "common.h"

template <int W>
__device__ __forceinline__ int64_t load_idx(const void* p, int64_t i) {
  if constexpr (W == LL_I32) return (int64_t)((const int32_t*)p)[i];
  else return ((const int64_t*)p)[i];
}
__device__ __forceinline__ int64_t load_idx_rt(const void* p, int64_t i, int w) {
  return w == LL_I32 ? (int64_t)((const int32_t*)p)[i] : ((const int64_t*)p)[i];
}

// One block per token; rows of [heads, hd] 16-bit elements copied as 16-byte vectors.
template <int VEC>
__global__ __launch_bounds__(256) void update_kv_buffer_kernel(
    const uint16_t* __restrict__ vals, const void* __restrict__ sel, uint16_t* __restrict__ buf,
    int heads, int hd, int64_t vst, int64_t vsh, int64_t bst, int64_t bsh, int idx_w) {
  const int64_t tok = blockIdx.x;
  const int64_t dst = load_idx_rt(sel, tok, idx_w);
  const int vph = hd / VEC;
  const int total = heads * vph;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / vph, j = (i % vph) * VEC;
    uint16_t v[VEC];
    VecIO<VEC>::load(vals + tok * vst + (int64_t)h * vsh + j, v);
    VecIO<VEC>::store(buf + dst * bst + (int64_t)h * bsh + j, v);
  }
}

extern "C" int ll_update_kv_buffer(const void* vals, const void* select_index, void* buf,
                                   int64_t tokens, int heads, int hd, int64_t v_stride_t,
                                   int64_t v_stride_h, int64_t b_stride_t, int64_t b_stride_h,
                                   int idx_width, void* stream) {
  if (idx_width != LL_I32 && idx_width != LL_I64) return LL_ERR_DTYPE;
  if (tokens < 0 || heads <= 0 || hd <= 0) return LL_ERR_SHAPE;
  if (tokens == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (hd % 8 == 0) && ll_aligned16(vals) && ll_aligned16(buf) && (v_stride_t % 8 == 0) &&
                   (v_stride_h % 8 == 0) && (b_stride_t % 8 == 0) && (b_stride_h % 8 == 0);
  if (vec)
    update_kv_buffer_kernel<8><<<dim3((unsigned)tokens), 256, 0, st>>>(
        (const uint16_t*)vals, select_index, (uint16_t*)buf, heads, hd, v_stride_t, v_stride_h,
        b_stride_t, b_stride_h, idx_width);
  else
    update_kv_buffer_kernel<1><<<dim3((unsigned)tokens), 256, 0, st>>>(
        (const uint16_t*)vals, select_index, (uint16_t*)buf, heads, hd, v_stride_t, v_stride_h,
        b_stride_t, b_stride_h, idx_width);
  return LL_LAUNCH_CHECK();
}

// fp8 KV cache (extension): the same scatter with the rows quantised to OCP e4m3 (gfx950's native fp8):
// head h < k_heads is divided by k_scale, the others by v_scale, clamped to +-448 and rounded to nearest even by
// v_cvt_pk_fp8_f32.  One block per token, 8 values per thread.
template <int DT>
__global__ __launch_bounds__(256) void update_kv_buffer_fp8_kernel(
    const uint16_t* __restrict__ vals, const void* __restrict__ sel, uint8_t* __restrict__ buf, int heads, int k_heads,
    int hd, int64_t vst, int64_t vsh, int64_t bst, int64_t bsh, float k_scale, float v_scale, int idx_w) {
  const int64_t tok = blockIdx.x;
  const int64_t dst = load_idx_rt(sel, tok, idx_w);
  const int vph = hd / 8;
  const int total = heads * vph;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / vph, j = (i % vph) * 8;
    uint16_t v[8];
    VecIO<8>::load(vals + tok * vst + (int64_t)h * vsh + j, v);
    const float sc = h < k_heads ? k_scale : v_scale;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(to_f32<DT>(v[e]) / sc, -448.0f), 448.0f);  // a true division: x / s, not x * (1 / s)
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *reinterpret_cast<uint2*>(buf + dst * bst + (int64_t)h * bsh + j) = uint2{(uint32_t)lo, (uint32_t)hi};
  }
}

extern "C" int ll_update_kv_buffer_fp8(const void* vals, const void* select_index, void* buf, int64_t tokens, int heads,
                                       int k_heads, int hd, int64_t v_stride_t, int64_t v_stride_h, int64_t b_stride_t,
                                       int64_t b_stride_h, float k_scale, float v_scale, int dtype, int idx_width,
                                       void* stream) {
  if (idx_width != LL_I32 && idx_width != LL_I64) return LL_ERR_DTYPE;
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (tokens < 0 || heads <= 0 || k_heads < 0 || k_heads > heads || hd <= 0 || hd % 8 != 0) return LL_ERR_SHAPE;
  if (!(k_scale > 0.f) || !(v_scale > 0.f)) return LL_ERR_ARG;
  if (tokens == 0) return LL_OK;
  if (!ll_aligned16(vals) || ((uintptr_t)buf & 7) || (v_stride_t % 8) || (v_stride_h % 8) || (b_stride_t % 8) || (b_stride_h % 8))
    return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16)
    update_kv_buffer_fp8_kernel<LL_F16><<<dim3((unsigned)tokens), 256, 0, st>>>(
        (const uint16_t*)vals, select_index, (uint8_t*)buf, heads, k_heads, hd, v_stride_t, v_stride_h, b_stride_t,
        b_stride_h, k_scale, v_scale, idx_width);
  else
    update_kv_buffer_fp8_kernel<LL_BF16><<<dim3((unsigned)tokens), 256, 0, st>>>(
        (const uint16_t*)vals, select_index, (uint8_t*)buf, heads, k_heads, hd, v_stride_t, v_stride_h, b_stride_t,
        b_stride_h, k_scale, v_scale, idx_width);
  return LL_LAUNCH_CHECK();
}

__global__ void update_kv_index_kernel(int32_t* __restrict__ table, const void* __restrict__ req,
                                       const void* __restrict__ seq, const void* __restrict__ sel,
                                       int64_t n, int64_t sb, int64_t ss, int rw, int sw, int lw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = load_idx_rt(req, i, rw);
  const int64_t l = load_idx_rt(seq, i, sw);
  table[r * sb + (l - 1) * ss] = (int32_t)load_idx_rt(sel, i, lw);
}

extern "C" int ll_update_kv_index(int32_t* table, const void* b_req_idx, const void* b_seq_len,
                                  const void* select_index, int64_t n, int64_t stride_b,
                                  int64_t stride_s, int req_width, int seq_width, int sel_width,
                                  void* stream) {
  if ((req_width | seq_width | sel_width) & ~1) return LL_ERR_DTYPE;
  if (n < 0) return LL_ERR_SHAPE;
  if (n == 0) return LL_OK;
  update_kv_index_kernel<<<dim3((unsigned)((n + 63) / 64)), 64, 0, (hipStream_t)stream>>>(
      table, b_req_idx, b_seq_len, select_index, n, stride_b, stride_s, req_width, seq_width, sel_width);
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// One launch for the per-token bookkeeping of a lockstep greedy decode step (the reference does it
// with a handful of host-issued tensor ops: model_runner.py:200-218 + llm_engine.py:173-213):
//   out[i, step] = next[i]; input_ids[i] = next[i]; positions[i] += 1;
//   cur_select_index[i] += batch (the bump allocator's next rows); b_seq_len[i] += 1;
//   table[b_req_idx[i], b_seq_len[i] - 1] = cur_select_index[i]   (= update_kv_index);  step += 1
// Single workgroup (rows are independent; the shared step counter is bumped after a barrier).
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void decode_advance_kernel(
    int64_t* __restrict__ out, int64_t out_stride, int64_t* __restrict__ step, const int64_t* __restrict__ next,
    int64_t* __restrict__ input_ids, int64_t* __restrict__ positions, int32_t* __restrict__ cur_select,
    int32_t* __restrict__ b_seq_len, const int32_t* __restrict__ b_req_idx, int32_t* __restrict__ table,
    int64_t t_sb, int64_t t_ss, int batch) {
  const int64_t st = *step;
  for (int i = threadIdx.x; i < batch; i += 256) {
    const int64_t tok = next[i];
    out[(int64_t)i * out_stride + st] = tok;
    input_ids[i] = tok;
    positions[i] += 1;
    const int32_t sel = cur_select[i] + batch;
    cur_select[i] = sel;
    const int32_t len = b_seq_len[i] + 1;
    b_seq_len[i] = len;
    table[(int64_t)b_req_idx[i] * t_sb + (int64_t)(len - 1) * t_ss] = sel;
  }
  __syncthreads();
  if (threadIdx.x == 0) *step = st + 1;
}

extern "C" int ll_decode_advance(int64_t* out, int64_t out_stride, int64_t* step, const int64_t* next_tokens,
                                 int64_t* input_ids, int64_t* positions, int32_t* cur_select_index,
                                 int32_t* b_seq_len, const int32_t* b_req_idx, int32_t* table,
                                 int64_t table_stride_b, int64_t table_stride_s, int batch, void* stream) {
  if (batch < 0) return LL_ERR_SHAPE;
  if (batch == 0) return LL_OK;
  if (!out || !step || !next_tokens || !input_ids || !positions || !cur_select_index || !b_seq_len ||
      !b_req_idx || !table)
    return LL_ERR_ARG;
  decode_advance_kernel<<<dim3(1), 256, 0, (hipStream_t)stream>>>(out, out_stride, step, next_tokens, input_ids,
                                                                 positions, cur_select_index, b_seq_len,
                                                                 b_req_idx, table, table_stride_b,
                                                                 table_stride_s, batch);
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// Continuous batching, steady state (executor/slot_batch.py:135-169 does this with ``+= 1`` and a
// gather against the slot table; here one launch, optionally also feeding the sampled tokens back):
//   b_seq_len[i] += 1;  cur_select_index[i] = table[b_req_idx[i], b_seq_len[i] - 1];
//   positions[i] = b_seq_len[i] - 1 (if given);  input_ids[i] = next_tokens[i] (if given)
// --------------------------------------------------------------------------- //
template <typename T>
__global__ __launch_bounds__(256) void slot_advance_kernel(
    T* __restrict__ b_seq_len, const T* __restrict__ b_req_idx, int32_t* __restrict__ cur_select,
    int64_t* __restrict__ positions, int64_t* __restrict__ input_ids, const int64_t* __restrict__ next,
    const int32_t* __restrict__ table, int64_t t_sb, int64_t t_ss, int batch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= batch) return;
  const int64_t len = (int64_t)b_seq_len[i] + 1;
  b_seq_len[i] = (T)len;
  cur_select[i] = table[(int64_t)b_req_idx[i] * t_sb + (len - 1) * t_ss];
  if (positions) positions[i] = len - 1;
  if (input_ids) input_ids[i] = next[i];
}

extern "C" int ll_slot_advance(void* b_seq_len, const void* b_req_idx, int32_t* cur_select_index,
                               int64_t* positions, int64_t* input_ids, const int64_t* next_tokens,
                               const int32_t* table, int64_t table_stride_b, int64_t table_stride_s,
                               int batch, int idx_width, void* stream) {
  if (idx_width != LL_I32 && idx_width != LL_I64) return LL_ERR_DTYPE;
  if (batch < 0) return LL_ERR_SHAPE;
  if (batch == 0) return LL_OK;
  if (!b_seq_len || !b_req_idx || !cur_select_index || !table || (input_ids && !next_tokens)) return LL_ERR_ARG;
  const dim3 grid((unsigned)((batch + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (idx_width == LL_I32)
    slot_advance_kernel<int32_t><<<grid, 256, 0, st>>>((int32_t*)b_seq_len, (const int32_t*)b_req_idx,
                                                       cur_select_index, positions, input_ids, next_tokens,
                                                       table, table_stride_b, table_stride_s, batch);
  else
    slot_advance_kernel<int64_t><<<grid, 256, 0, st>>>((int64_t*)b_seq_len, (const int64_t*)b_req_idx,
                                                       cur_select_index, positions, input_ids, next_tokens,
                                                       table, table_stride_b, table_stride_s, batch);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_abi_version(void) { return 1; }
