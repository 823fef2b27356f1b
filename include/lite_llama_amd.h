/*
 * lite_llama_amd -- C ABI of the MI355X (gfx950) decode hot path.
 *
 * Drop-in boundary for the kernel layer of harleyszhang/lite_llama: every entry
 * point below replaces one Triton host wrapper of the reference (cited per
 * function as  lite_llama/kernels/<file>:<lines>).  The reference is pure
 * Python + Triton, so the binding a maintainer adds is a ctypes stub (shown in
 * INTEGRATION.md); the shipped Python mirror lives in lite_llama_amd/kernels/.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers unless the name says host;
 *   - strides are in ELEMENTS, sizes in elements unless suffixed _bytes;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *   - asynchronous, NO allocation and no device synchronisation -> every call is
 *     hipGraph-capturable; the library keeps no mutable state of its own: scratch
 *     (`workspace`, `counters`) is passed in by the caller (sizes from the
 *     *_workspace helpers).  Calls are re-entrant as long as two calls that may be
 *     in flight at the same time (different streams / threads) are given DIFFERENT
 *     scratch buffers: the split-K slabs and merge counters are written by the
 *     kernels.  The Python mirror keeps one scratch set per (device, stream);
 *   - return 0 on success, a negative LL_ERR_* code on an argument error (the
 *     Python mirror raises the same exception types the reference raises);
 *     kernels themselves never report errors (same as the reference).
 */
#ifndef LITE_LLAMA_AMD_H
#define LITE_LLAMA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types */
#define LL_F16 0
#define LL_BF16 1
#define LL_F32 2
/* index widths (reference passes int32 or int64 index tensors) */
#define LL_I32 0
#define LL_I64 1
/* 8-bit weight formats (w8a16.py:155-216, fused_moe.py:36-38) */
#define LL_W_F16 0
#define LL_W_FP8E4M3 1
#define LL_W_INT8 2

#define LL_OK 0
#define LL_ERR_DTYPE (-1)
#define LL_ERR_SHAPE (-2)
#define LL_ERR_ARG (-3)
#define LL_ERR_LAUNCH (-4)

/* Library / ABI version; bumped on any signature change. */
int ll_abi_version(void);

/* ---- a1: skip_rmsnorm  (kernels/skip_rmsnorm.py:192-234) -------------------
 * s = x + r (fp32); r <- s rounded (IN PLACE, when residual != NULL);
 * var = sum(s*s/n); y = (s*rsqrt(var+eps)).to(dtype) * w.  rows x n, contiguous. */
int ll_skip_rmsnorm(void* y, const void* x, void* residual, const void* weight,
                    int64_t rows, int64_t n, float eps, int dtype, void* stream);

/* ---- a7: swiglu_forward  (kernels/swiglu.py:45-65) --------------------------
 * c = silu(a.f32) * b, rows x n, contiguous. */
int ll_swiglu(void* c, const void* a, const void* b, int64_t rows, int64_t n, int dtype,
              void* stream);

/* ---- element-wise activations  (kernels/activations.py:19-57: relu, leaky_relu, tanh, gelu) ----
 * y[i] = f(x[i].f32) rounded to the storage dtype; kind 0 relu, 1 leaky_relu (slope 0.01 in the storage dtype),
 * 2 tanh, 3 gelu (erf form); n contiguous elements, y may alias x. */
int ll_activation(void* y, const void* x, int64_t n, int kind, int dtype, void* stream);

/* ---- a2: rope_emb_forward  (kernels/rope_emb.py:86-134) ---------------------
 * In-place half-split rotation of q [tokens, n_qh, hd] and k [tokens, n_kh, hd]
 * (head and dim contiguous, row strides given).  Token t reads
 * cos[t / seq_len, t % seq_len, : hd/2].  cs_dtype is the table dtype. */
int ll_rope(void* q, void* k, const void* cos_t, const void* sin_t, int64_t tokens, int n_qh,
            int n_kh, int hd, int64_t q_row_stride, int64_t k_row_stride, int64_t seq_len,
            int64_t cos_b_stride, int64_t cos_s_stride, int64_t sin_b_stride, int64_t sin_s_stride,
            int qk_dtype, int cs_dtype, void* stream);
/* Decode-step fusion of the two entries above (no reference counterpart: the reference launches
 * rope_emb.py:86-134 and update_kv_buffer.py:54-89 back to back): rotate q and the K heads of
 * kv = [tokens, 2*n_kh, hd] in place and scatter the rotated K heads + the V heads of token i to
 * kv_buffer row select_index[i].  Bit-identical to ll_rope followed by ll_update_kv_buffer.
 * positions (int64 [tokens], optional): cos/sin are then position-indexed tables [max_pos, >= hd/2]
 * (row stride = the *_s_stride arguments) and token i reads row positions[i] -- the rotary producer
 * (models/rotary_embedding.py:34-137) leaves the per-step graph. */
int ll_rope_kv_update(void* q, void* kv, const void* cos_t, const void* sin_t, void* kv_buffer,
                      const void* select_index, int64_t tokens, int n_qh, int n_kh, int hd,
                      int64_t q_row_stride, int64_t kv_row_stride, int64_t seq_len,
                      int64_t cos_b_stride, int64_t cos_s_stride, int64_t sin_b_stride,
                      int64_t sin_s_stride, int64_t pool_stride_t, int64_t pool_stride_h,
                      int qk_dtype, int cs_dtype, int idx_width, const int64_t* positions,
                      void* stream);

/* ---- a3: update_kv_buffer  (kernels/update_kv_buffer.py:54-89) ---------------
 * buf[idx[i], h, :] = vals[i, h, :]  (bit-exact copy; 2-byte elements). */
int ll_update_kv_buffer(const void* vals, const void* select_index, void* buf, int64_t tokens,
                        int heads, int hd, int64_t v_stride_t, int64_t v_stride_h,
                        int64_t b_stride_t, int64_t b_stride_h, int idx_width, void* stream);

/* ---- a4: update_kv_index  (kernels/update_kv_index.py:50-88) -----------------
 * table[b_req_idx[i], b_seq_len[i]-1] = select_index[i]; table is int32. */
int ll_update_kv_index(int32_t* table, const void* b_req_idx, const void* b_seq_len,
                       const void* select_index, int64_t n, int64_t stride_b, int64_t stride_s,
                       int req_width, int seq_width, int sel_width, void* stream);

/* ---- a5: flash_decoding  (kernels/flashdecoding.py:316-380) ------------------
 * out[b,h,:] = softmax(q[b,h] . K[rows]^T * scale) V[rows],
 * rows = table[b_req_idx[b], :b_seq_len[b]], kv head = h / (hq/hkv).
 * K/V are (possibly strided) views [tokens, hkv, d], d contiguous.  Scratch:
 * mid_o fp32 [batch, hq, nparts, d], mid_lse fp32 [batch, hq, nparts], with
 * nparts = ll_flash_decoding_num_partitions(max_len).
 * counters: NULL, or an int32 vector of batch * hkv * ceil((hq/hkv)/16) entries that is ZERO at
 * launch; with it the log-sum-exp merge runs inside the same launch (the wave finishing a row's
 * last partition merges; the vector is zero again when the kernel ends); mid_lse must then hold
 * 32 floats per (row, head, partition) -- batch * hq * nparts * 32 -- one cache line per record.
 * Same values either way. */
int ll_flash_decoding_num_partitions(int64_t max_len);
/* Waves of the one-workgroup-per-(row, KV head) decode form for contexts of up to max_len tokens launched as `workgroups` = batch x
 * KV heads (x head groups) workgroups (2 .. 8 partitions: a wave per partition; more: 8 waves that walk the partitions, when the
 * launch has >= 128 workgroups); 0: the form does not apply -- ll_decode_attention_partials needs it. */
int ll_flash_decoding_group_waves(int64_t max_len, int64_t workgroups);
int ll_flash_decoding(void* out, const void* q, const void* k_cache, const void* v_cache,
                      const int32_t* table, const void* b_req_idx, const void* b_seq_len,
                      float* mid_o, float* mid_lse, int batch, int hq, int hkv, int d,
                      int64_t max_len, float qk_scale, int64_t q_stride_b, int64_t q_stride_h,
                      int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t,
                      int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                      int64_t table_stride_b, int dtype, int req_width, int seq_width,
                      int32_t* counters, void* stream);

/* Decode-step attention in one launch (executor extension): rope of q and of the new token's K
 * with position-indexed tables (kernels/rope_emb.py:86-134 arithmetic), the scatter of that K and V
 * into pool row select_index[b] (kernels/update_kv_buffer.py:54-89) and flash_decoding, with the same
 * values as ll_rope_kv_update followed by ll_flash_decoding.  q [batch, hq, d] is read un-rotated and
 * not written back; kv_new [batch, 2*hkv, d] (K heads first, row stride given); cos/sin 16-bit
 * tables [max_pos, >= d/2] of the q dtype (row stride given); positions int64 [batch].  k_cache /
 * v_cache are the (writable) pool views.  counters / mid_lse as for ll_flash_decoding's one-launch
 * form (required).  d >= 64 and hq/hkv <= 16, else LL_ERR_SHAPE; the select rows must be distinct
 * and already named by the table at position b_seq_len[b]-1.
 * q_norm_weight / k_norm_weight ([d], q dtype; both or both NULL; d == 128): Qwen3's per-head RMSNorm of q and of the new
 * K heads (models/qwen3.py q_norm / k_norm; eps = norm_eps) applied in front of the rotation, with the values
 * ll_skip_rmsnorm gives on the [.., d] views. */
int ll_decode_attention(void* out, const void* q, const void* kv_new, int64_t kv_row_stride,
                        const void* cos_t, const void* sin_t, int64_t cs_row_stride,
                        const int64_t* positions, const void* select_index, int sel_width,
                        void* k_cache, void* v_cache, const int32_t* table, const void* b_req_idx,
                        const void* b_seq_len, float* mid_o, float* mid_lse, int batch, int hq, int hkv,
                        int d, int64_t max_len, float qk_scale, int64_t q_stride_b, int64_t q_stride_h,
                        int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                        int64_t o_stride_b, int64_t o_stride_h, int64_t table_stride_b, int dtype,
                        int req_width, int seq_width, int32_t* counters, const void* q_norm_weight,
                        const void* k_norm_weight, float norm_eps, void* stream);

/* ll_decode_attention whose q / k_new / v_new arrive as the fp32 split-K partials [s_count][batch][(hq + 2 hkv) * d] of the
 * fused q|k|v projection (ll_w4a16_matmul_prepacked epilogue 2) plus the projection bias (or NULL): every workgroup adds
 * the partials of its (row, KV head), rounds once to the pool dtype -- the value the projection would have stored -- and
 * continues as ll_decode_attention.  Contexts of 2..8 partitions (129..1024 tokens); LL_ERR_SHAPE otherwise.
 * int32_a_scale [batch] / int32_w_scale [row width] (both or both NULL; fp16, d == 128): the planes are the exact int32 sums
 * of a smoothquant projection (ll_dense_partials wfmt 3); the value summed is fp16(((float)sum * a_scale[row]) * w_scale[col]
 * (+ bias)), what ll_w8a8_matmul stores. */
int ll_decode_attention_partials(void* out, const float* qkv_partials, int s_count, const void* qkv_bias,
                                 const void* cos_t, const void* sin_t, int64_t cs_row_stride, const int64_t* positions,
                                 const void* select_index, int sel_width, void* k_cache, void* v_cache,
                                 const int32_t* table, const void* b_req_idx, const void* b_seq_len, int batch, int hq,
                                 int hkv, int d, int64_t max_len, float qk_scale, int64_t k_stride_t, int64_t k_stride_h,
                                 int64_t v_stride_t, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                 int64_t table_stride_b, int dtype, int req_width, int seq_width,
                                 const void* q_norm_weight, const void* k_norm_weight, float norm_eps,
                                 const float* int32_a_scale, const float* int32_w_scale, void* stream);

/* ---- a6: flash_attention2_no_pad  (kernels/flashattention2_nopad.py:175-231) --
 * Varlen causal prefill over freshly projected q/k/v (exp2 softmax; sm_scale
 * already carries log2 e).  q/o [tokens, hq, d], k/v [tokens, hkv, d]. */
int ll_flash_attention_nopad(void* out, const void* q, const void* k, const void* v,
                             const void* b_start_loc, const void* b_seq_len, int batch, int hq,
                             int hkv, int d, int64_t max_seq_len, float sm_scale,
                             int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t,
                             int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                             int64_t o_stride_t, int64_t o_stride_h, int dtype, int start_width,
                             int seq_width, void* stream);

/* skip_rmsnorm over a projection left as S fp32 split-K partials [S][rows][n] (see ll_w4a16_partials_count):
 * x = fp16(sum_s partials[s]) -- the value the projection would have stored -- then exactly ll_skip_rmsnorm
 * with that x (residual required, updated in place). */
int ll_skip_rmsnorm_partials(void* y, const float* partials, int s_count, void* residual, const void* weight,
                             int64_t rows, int64_t n, float eps, int dtype, void* stream);
/* skip_rmsnorm over the fused-MoE block's per-slot rows [rows][k_count][n] (16-bit, router weights folded in): x = the value
 * ll_moe_sum gives (fp32 sum over the k slots in order, one rounding; fused_moe.py:318-335), then exactly ll_skip_rmsnorm
 * (residual required, updated in place) -- the moe_sum launch and the norm launch as one.  k_count <= 8, n % 8 == 0, n <= 8192. */
int ll_skip_rmsnorm_slots(void* y, const void* slots, int k_count, void* residual, const void* weight, int64_t rows, int64_t n,
                          float eps, int dtype, void* stream);

/* ---- a8: w4a16_matmul  (kernels/quantization/w4a16.py:152-207) ---------------
 * out[m,n] = sum_k x[m,k] * ((nib(qweight[n,k/8],k%8) - zeros[n,k/g]) * scales[n,k/g]) (+bias)
 * x fp16 [M,K] (row stride given), qweight int32 [N,K/8], scales/zeros fp32
 * [N,K/g] (row stride given), bias fp16 [N] or NULL, out fp16 [M,N] contiguous.
 * `workspace` (fp32) and `counters` (int32, must be zero on entry; left zero on
 * exit) are stream-K scratch sized by ll_gemm_workspace(m, n, k). */
int ll_gemm_workspace(int64_t m, int64_t n, int64_t k, int64_t* workspace_floats,
                      int64_t* counter_ints);
int ll_w4a16_matmul(void* out, const void* x, const int32_t* qweight, const float* scales,
                    const float* zeros, const void* bias, int64_t m, int64_t n, int64_t k,
                    int group_size, int64_t x_stride_m, int64_t qw_stride_n, int64_t s_stride_n,
                    float* workspace, int32_t* counters, void* stream);
/* Load-time companion (no reference counterpart; the reference keeps the fp32 grids only):
 * packed[g][n] = 8 bytes {fp16 s, fp16 s, fp16 -z*s, fp16 -z*s} -- the pair the decode engine (ll_w4a16_matmul_prepacked)
 * streams, transposed so that one 128-row tile of one group is contiguous. */
int ll_w4a16_pack_scales(void* packed, const float* scales, const float* zeros, int64_t n,
                         int64_t groups, int64_t s_stride_n, void* stream);

/* Load-time weight layout of the decode engine (no reference counterpart; the reference reserves
 * lite_llama/models/quantization/_layout/__init__.py:1-6 for exactly this): a bit-exact permutation of
 * qweight [N, K/8] into the order the M <= 64 kernel streams it (per (128-row tile, 128-k chunk) 8 KB =
 * [wave 8][lane 64] x 16 B; inside a word even nibbles first).  packed: n * k / 2 bytes, n % 128 == 0,
 * k % 128 == 0.  ll_w4a16_matmul_prepacked consumes it together with ll_w4a16_pack_scales' output
 * (group_size = 128 * 2^j); epilogue bit 0: 0 = w4a16_matmul, 1 = the gate/up + swiglu fusion (no reference counterpart: the
 * reference runs gate_proj, up_proj and swiglu.py:45-65 as three launches; weight rows interleaved -- row 2j = gate_j, row
 * 2j+1 = up_j, n = 2 * intermediate -- out [M, n/2] = silu(gate) * up with the stand-alone kernels' rounding); bits 8-9
 * (tests / tuning): 0 = tile width chosen by the host plan, 1 / 2 = force 128- / 256-row tiles (256 needs
 * n % 256 == 0).  Same arithmetic as ll_w4a16_matmul (only the fp32 summation order differs). */
int ll_w4a16_pack_weights(void* packed, const int32_t* qweight, int64_t n, int64_t k, int64_t qw_stride_n,
                          void* stream);
/* The inverse (bit-exact): packed -> qweight [N, K/8] in the reference layout (models/quantization/params/int4.py:33-49), so
 * that the load-time layout can be the only resident copy of the weights. */
int ll_w4a16_unpack_weights(int32_t* qweight, const void* packed, int64_t n, int64_t k, int64_t qw_stride_n,
                            void* stream);
int ll_w4a16_prepacked_supported(int64_t m, int64_t n, int64_t k, int group_size); /* 1 / 0 */
/* epilogue 2 of ll_w4a16_matmul_prepacked = split-K partial mode (decode-step fusion, no reference counterpart):
 * `out` is an fp32 [S][M][N] buffer, S = ll_w4a16_partials_count(...) (0 = shape not served: use epilogue 0);
 * the S partial sums are added by the consumer of the projection, ll_skip_rmsnorm_partials below -- the
 * GEMM then has no cross-workgroup merge at all (bias must be NULL). */
int ll_w4a16_partials_count(int64_t m, int64_t n, int64_t k, int group_size);
/* ... for a launch that will be given `epilogue` (2, or 2 | 0x100 = the unit loop forced: tests / tuning -- its k-split differs from
 * the short-stream engine's, which takes the launches of a few tens of KB per CU since round 6, gemm_short.hip). */
int ll_w4a16_partials_count_ex(int64_t m, int64_t n, int64_t k, int group_size, int epilogue);
/* Host-side introspection of the short-stream engine (no device work): out8 = [takes the split-K partial launch 0 / 1, grid, row
 * groups per work item R, k-slices S (= planes), pieces per consumer wave (template bound), 64-k blocks per slice, slices with one
 * block more, LDS bytes]. */
int ll_w4a16_short_plan(int64_t m, int64_t n, int64_t k, int group_size, int32_t* out8);
/* ... and of its all-of-K form for narrow FINISHED outputs (gemm_short_full.hip; ll_w4a16_matmul_prepacked routes epilogues 0 / 1
 * there when the row-group engine does not take the launch): out8 = [takes it 0 / 1, work items (= grid), row groups per item R,
 * 32-row batch tiles per item, batch halves, pieces per consumer wave (template bound), ring slots, LDS bytes].
 * Semantics: kernels/quantization/w4a16.py:152-207 (+ bias), kernels/swiglu.py:45-65 for the fused gate|up. */
int ll_w4a16_short_full_plan(int64_t m, int64_t n, int64_t k, int group_size, int32_t* out8);
int ll_w4a16_matmul_prepacked(void* out, const void* x, const void* wpacked, const void* spacked,
                              const void* bias, int64_t m, int64_t n, int64_t k, int group_size,
                              int64_t x_stride_m, float* workspace, int32_t* counters, int epilogue,
                              void* stream);
/* Host-side introspection of the row-group engine (gemm_w4_v4.hip, round 5) for a finished-output launch (epilogue 0 / 1): out8 =
 * [takes the launch 0 / 1, grid, row groups per workgroup, rbase, rrem, compute units assumed, LDS bytes, 0]; workgroup t owns the
 * 32-row groups [t * rbase + min(t, rrem), + rbase + (t < rrem)).  No device work. */
int ll_w4a16_v4_plan(int64_t m, int64_t n, int64_t k, int group_size, int epilogue, int32_t* out8);

/* The same product for MANY rows (prefill: m = batch x prompt tokens; w4a16.py:152-207 tiles M x N) over the same load-time
 * layouts: 256 x 256 x 64 MFMA tiles, the weight tile dequantised once per 256 rows (gemm_w4_prefill.hip).  epilogue 0: out
 * [m][n] (+ bias); 1: fused gate|up rows interleaved -> out [m][n / 2] = silu(gate) * up.  n % 256 == 0, k % 128 == 0. */
int ll_w4a16_mtiled_supported(int64_t m, int64_t n, int64_t k, int group_size); /* 1 / 0 */
int ll_w4a16_matmul_prepacked_mtiled(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias,
                                     int64_t m, int64_t n, int64_t k, int group_size, int64_t x_stride_m, int epilogue,
                                     void* stream);
/* Host-side introspection (tests, DESIGN_NOTEBOOK.md; no device work): the launch plan ll_w4a16_matmul_prepacked uses for (n, k,
 * epilogue) as 16 ints -- grid, 128-row blocks per tile, tiles, chunks, slab slots, tile-group split (gt, gbase, grem, lead,
 * xcd_shift), stream-K units per workgroup, four reserved zeros, compute units assumed.  tests/test_host_cpu.py restates the
 * kernel's per-workgroup decode on it. */
int ll_w4a16_v3_plan(int64_t m, int64_t n, int64_t k, int group_size, int epilogue, int32_t* out16);
/* ---- a9: w8a16_matmul  (kernels/quantization/w8a16.py:155-216) ---------------
 * qweight [N,K] uint8 (fp8-e4m3 bits) or int8; scales fp32 [ceil(N/gn), ceil(K/gk)]. */
int ll_w8a16_matmul(void* out, const void* x, const void* qweight, const float* scales,
                    const void* bias, int64_t m, int64_t n, int64_t k, int group_n, int64_t group_k,
                    int wfmt, int64_t x_stride_m, int64_t qw_stride_n, int64_t s_stride_n,
                    int64_t s_stride_k, float* workspace, int32_t* counters, void* stream);

/* ---- a10: smoothquant_matmul  (kernels/quantization/w8a8.py:151-217) ----------
 * step 1: per-row scale = absmax/127 (1.0 if 0), q = trunc(x/scale) int8;
 * step 2: int8 x int8 -> int32 (exact) -> * a_scale[m] * w_scale[n] (+bias) -> fp16. */
int ll_quantize_activations_int8(int8_t* q, float* a_scale, const void* x, int64_t m, int64_t k,
                                 int64_t x_stride_m, void* stream);
int ll_w8a8_matmul(void* out, const int8_t* qa, const float* a_scale, const int8_t* qweight,
                   const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                   int64_t qw_stride_n, int32_t* acc_out /* nullable: raw int32 accumulators [M,N] */,
                   int32_t* workspace, int32_t* counters, void* stream);

/* Calls of more than 64 rows (prefill) of the two entries above take an M x N tiled MFMA GEMM (gemm_w8_prefill.hip: 256 x 256
 * tiles, the weight tile fetched once per 256 rows -- the reference's kernels tile M x N the same way, w8a16.py:155-216,
 * w8a8.py:151-217); this predicate says whether a shape is served (1 / 0; otherwise the 64-row weight-streaming tile loops
 * over M).  wfmt: 1 fp8 e4m3 / 2 int8 weights under fp16 activations, 3 int8 x int8.  n % 32 == 0; k % 64 (128 for wfmt 3). */
int ll_w8_mtiled_supported(int64_t m, int64_t n, int64_t k, int wfmt, int64_t group_k);

/* Split-K partial mode of the decode-shaped 8-bit / 16-bit engine (no reference counterpart; the int4 engine's epilogue 2 for
 * the other formats): partials [S][M][N] fp32 with the block scales applied, summed by the projection's consumer
 * (ll_skip_rmsnorm_partials, ll_decode_attention_partials) -- no finish launch.  wfmt: 1 fp8 e4m3 / 2 int8 (fp16
 * activations; scales as in ll_w8a16_matmul) / 4 fp16 / 5 bf16 weights (activations of the weight's type, scales NULL;
 * w_stride in elements) / 3 = smoothquant (x int8 rows quantised by the caller, w int8, scales NULL: the planes are exact
 * int32 sums, consumed with the per-token / per-channel scales by ll_skip_rmsnorm_q8 / ll_w8a8_finish_swiglu).
 * ll_dense_partials_count: S for (shape, format, cap) -- 0 = not served; ll_dense_partials returns
 * the S it wrote (the same number), 0 when alignment rules decline the call, < 0 on error. */
int ll_dense_partials_count(int64_t m, int64_t n, int64_t k, int wfmt, int max_splits);
int ll_dense_partials(float* partials, const void* x, const void* w, const float* scales, int64_t m, int64_t n, int64_t k,
                      int group_n, int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride, int64_t s_stride_n,
                      int64_t s_stride_k, int max_splits, void* stream);

/* Decode-step fusions around smoothquant_matmul (csrc/w8a8_fused.hip; reference kernels/quantization/w8a8.py:28-217 + the
 * skip_rmsnorm / swiglu launches between two W8A8 projections).
 * ll_skip_rmsnorm_q8: skip_rmsnorm (fp16) whose input is EITHER x [rows][n] (planes NULL) OR the int32 planes
 * [s_count][rows][n] of ll_dense_partials(wfmt 3) with a_scale [rows], w_scale [n], optional bias [n] (x NULL): the
 * normalised input is then fp16(((float)sum * a_scale[m]) * w_scale[n] (+ bias)) -- what ll_w8a8_matmul stores.
 * residual (updated in place) or NULL.  Outputs, each optional (not both NULL): y fp16 [rows][n]; q int8 [rows][n] +
 * q_scale [rows] = ll_quantize_activations_int8(y).  Bit-identical to the separate launches.  n % 8 == 0, n <= 8192.
 * ll_w8a8_finish_swiglu: out [m][n/2] = silu(gate) * up over the planes of a fused gate|up projection with interleaved
 * rows (gate_j, up_j), gate / up finished as above. */
int ll_skip_rmsnorm_q8(void* y, int8_t* q, float* q_scale, const void* x, const int32_t* planes, int s_count,
                       const float* a_scale, const float* w_scale, const void* bias, void* residual, const void* weight,
                       int64_t rows, int64_t n, float eps, void* stream);
int ll_w8a8_finish_swiglu(void* out, const int32_t* planes, int s_count, const float* a_scale, const float* w_scale,
                          int64_t m, int64_t n, void* stream);
/* (internal dispatch target of ll_quantize_activations_int8: the row-in-registers quantiser; 1 launched / 0 declined) */
int ll_quant_act_cached_try(int8_t* q, float* a_scale, const void* x, int64_t m, int64_t k, int64_t x_stride_m, void* stream);

/* ---- unquantised 16-bit linears at decode shapes (models/quantization/methods/unquantized.py:21-22 and the lm_head of
 * models/base.py:486-489: torch F.linear, i.e. a vendor GEMM, in the reference) -------------------------------------------
 * out[m, n] = x[m, :] . w[n, :] (+ bias[n]); x [m, k] (row stride x_stride elements), w [n, k] (row stride w_stride), bias,
 * out [m, n] all fp16 or all bf16; fp32 accumulation, one rounding.  Split-K weight streaming (gemm_w8_skinny.hip without
 * the dequantisation): `partials` = ll_gemm_workspace(m, n, k) floats of scratch.  Serves m <= 64, k % 64 == 0, n % 4 == 0,
 * 16-byte aligned rows; returns 1 when the launches were issued, 0 when the shape is not served (the caller keeps its
 * library GEMM), < 0 on error. */
int ll_dense16_matmul(void* out, const void* x, const void* w, const void* bias, int64_t m, int64_t n, int64_t k,
                      int64_t x_stride, int64_t w_stride, int dtype, void* partials, void* stream);
/* Round 5: the same product on the row-group loop (gemm_w16_rows.hip) -- every workgroup owns a run of 32-row groups and all of K,
 * weights and activations streamed through an LDS ring by LDS-DMA, finished outputs only (no planes, no finish launch).
 * epilogue 0: out [m][n] (+ bias); 1: the rows of w are (gate_j, up_j) pairs -> out [m][n / 2] = silu(gate) * up (swiglu_forward's
 * arithmetic on the rounded outputs).  m <= 64, n % 32 == 0, k % 128 == 0, element strides % 8 == 0. */
int ll_dense16_rows_supported(int64_t m, int64_t n, int64_t k, int epilogue); /* 1 / 0 */
int ll_dense16_rows_matmul(void* out, const void* x, const void* w, const void* bias, int64_t m, int64_t n, int64_t k,
                           int64_t x_stride_m, int64_t w_stride_n, int dtype, int epilogue, void* stream);
/* The same loop for a SmoothQuant (W8A8) projection whose rows arrive already quantised: exact int32 sums on mfma_i32_32x32x32_i8,
 * then ll_w8a8_matmul's epilogue fp16(((float)sum * a_scale[m]) * w_scale[n] (+ bias)) (kernels/quantization/w8a8.py:118-149);
 * epilogue 1: rows of qw interleaved (gate_j, up_j) -> out [m][n / 2] = silu(gate) * up (= ll_w8a8_finish_swiglu).  One launch,
 * no planes.  m <= 64, n % 32 == 0, k % 256 == 0, byte strides % 16 == 0. */
int ll_w8a8_rows_supported(int64_t m, int64_t n, int64_t k, int epilogue); /* 1 / 0 */
int ll_w8a8_rows_matmul(void* out, const int8_t* qa, const float* a_scale, const int8_t* qw, const float* w_scale,
                        const void* bias, int64_t m, int64_t n, int64_t k, int64_t qa_stride_m, int64_t qw_stride_n,
                        int epilogue, void* stream);

/* ---- a11: fused_moe pieces  (kernels/fused_moe.py:45-99, :236-292, :298-335) ---
 * moe_align_block_size: sorted_ids int32[num_slots + E*(block-1)] (sentinel =
 * num_slots), expert_ids int32[ceil(max_padded/block)], num_post int32[1]. */
int ll_moe_align_block_size(const void* topk_ids, int ids_width, int64_t num_slots, int num_experts,
                            int block_size, int32_t* sorted_ids, int32_t* expert_ids,
                            int32_t* num_post, void* stream);
/* The same outputs, bit for bit, for prefill-sized inputs (more than 1024 slots): three launches -- per-chunk counts, prefix
 * over chunks and experts, stable placement -- over `workspace` (int32 words, ll_moe_align_workspace_ints of them; contents
 * undefined before and after).  ll_moe_align_workspace_ints returns 0 for inputs the one-workgroup kernels serve;
 * ll_moe_align_block_size_ws falls back to ll_moe_align_block_size when the workspace is null or too small
 * (kernels/fused_moe.py:45-99: the reference sorts on the host side of Triton with torch ops at any size). */
int64_t ll_moe_align_workspace_ints(int64_t num_slots, int num_experts);
int ll_moe_align_block_size_ws(const void* topk_ids, int ids_width, int64_t num_slots, int num_experts, int block_size,
                               int32_t* sorted_ids, int32_t* expert_ids, int32_t* num_post, int32_t* workspace,
                               int64_t workspace_ints, void* stream);
/* c[slot, :] = (a[slot / top_k, :] @ w[expert(slot)].T) (* topk_w[slot]).  mul_routed_weight: bit 0 = multiply by topk_w[slot]
 * (fp32, fused_moe.py:203-205); bit 1 (extension) = the rows of w are (gate_j, up_j) pairs -- a load-time interleave of the
 * stacked gate|up matrix -- and c is [num_slots, n / 2] = silu(gate) * up on the fp16-rounded GEMM outputs: the values of
 * ll_silu_and_mul over the [num_slots, n] tensor the two-launch form stores (fused_moe.py:298-315), one launch less. */
int ll_moe_gemm(void* c, const void* a, const void* w, const float* w_scale, const void* topk_w,
                const int32_t* sorted_ids, const int32_t* expert_ids, const int32_t* num_post,
                int64_t num_slots, int64_t em, int block_m, int64_t n, int64_t k, int top_k,
                int mul_routed_weight, int wfmt, int group_n, int64_t group_k, int64_t a_stride_m,
                int64_t w_stride_e, int64_t w_stride_n, int64_t s_stride_e, int64_t s_stride_n,
                int64_t s_stride_k, int dtype, void* stream);
int ll_silu_and_mul(void* out, const void* x, int64_t rows, int64_t n, int dtype, void* stream);
/* out[r, j] = silu(x[r, 2 j]) * x[r, 2 j + 1] over x [rows, 2 n]: the fused gate|up output of row-INTERLEAVED weights (round 5). */
int ll_silu_and_mul_pairs(void* out, const void* x, int64_t rows, int64_t n, int dtype, void* stream);
int ll_moe_sum(void* out, const void* x, int64_t tokens, int top_k, int64_t n, int dtype,
               void* stream);
/* Router tail (models/qwen3_moe.py:85-100 after the router GEMM): probabilities = softmax over ALL experts in fp32 of
 * the 16-bit logits [tokens, experts] (row stride logits_stride), the top_k largest in descending order (equal values:
 * lower expert first), optionally renormalised to sum 1, cast to the logits' dtype -> weights_out [tokens, top_k];
 * ids_out int64 [tokens, top_k].  experts <= 1024, top_k <= 64. */
int ll_moe_route_topk(void* weights_out, int64_t* ids_out, const void* logits, int64_t tokens, int experts,
                      int64_t logits_stride, int top_k, int norm_topk_prob, int dtype, void* stream);
/* The WHOLE router of a decode batch in-tree (models/qwen3_moe.py:102-111: the unquantised fp16 gate linear + the tail above):
 * x [tokens, hidden] (row stride x_stride) times gate_w [experts, hidden] (row stride w_stride) as ceil(hidden / 512) split-K fp32
 * planes in `planes` (ll_moe_router_workspace_floats fp32 of scratch, no zeroing), then the tail reading a token's logits as the
 * planes' sum rounded once to the activation dtype (what the reference's 16-bit GEMM stores).  tokens <= 64, experts % 32 == 0
 * (<= 1024), hidden % 128 == 0: ll_moe_router_supported says so (1 / 0); other shapes keep GEMM + ll_moe_route_topk. */
int ll_moe_router_supported(int64_t tokens, int experts, int64_t hidden);
int64_t ll_moe_router_workspace_floats(int64_t tokens, int experts, int64_t hidden);
/* align_block > 0: also moe_align_block_size(ids_out, align_block, experts) -> sorted_ids / expert_ids / num_post (sized as for
 * ll_moe_align_block_size) -- inside the tail's launch (the last workgroup to arrive at `counter`, one int32 that is zero before
 * the call and left zero) when the decode-sized align body fits a workgroup, as a following launch otherwise.  0: the four
 * pointers are ignored. */
int ll_moe_router(void* weights_out, int64_t* ids_out, const void* x, const void* gate_w, float* planes, int64_t tokens,
                  int experts, int64_t hidden, int64_t x_stride, int64_t w_stride, int top_k, int norm_topk_prob, int dtype,
                  int align_block, int32_t* sorted_ids, int32_t* expert_ids, int32_t* num_post, int32_t* counter, void* stream);

/* Decode-step bookkeeping in one launch (executor extension; the reference issues these as
 * separate tensor ops, model_runner.py:200-218 + llm_engine.py:173-213): records the sampled
 * tokens, feeds them back, advances positions / sequence lengths / the bump-allocated KV rows and
 * writes the new rows into the token table (= update_kv_index).  int64: out [batch, *] (row stride
 * given), step [1], next_tokens, input_ids, positions; int32: cur_select_index, b_seq_len,
 * b_req_idx, table. */
int ll_decode_advance(int64_t* out, int64_t out_stride, int64_t* step, const int64_t* next_tokens,
                      int64_t* input_ids, int64_t* positions, int32_t* cur_select_index,
                      int32_t* b_seq_len, const int32_t* b_req_idx, int32_t* table,
                      int64_t table_stride_b, int64_t table_stride_s, int batch, void* stream);

/* ---- device-side KV row allocator -------------------------------------------
 * Replaces the search of executor/kv_cache_manager.py:219-267 (``nonzero`` over the use-count
 * vector + two ``.item()`` reads = three device synchronisations per call) with three stream-ordered
 * launches and no read-back.  state: int32 use count per row [n_rows]; free_rows: int64 [1] device
 * counter of zero-count rows (debited by ``need`` on success); decision: int64 [2] = {mode, start}
 * with mode 0 = nothing allocated (fewer than ``need`` free rows, or no run in contiguous-only mode),
 * 1 = contiguous run starting at ``start``, 2 = the first ``need`` free rows in ascending order;
 * out_index: int32 [need] row ids (untouched when mode 0).  contiguous_first: 0 = scattered only
 * (alloc_kvcache :219-231), 1 = first run of ``need`` consecutive free rows else scattered
 * (alloc_kvcache_index :270-299 general path), 2 = contiguous only (alloc_contiguous_kvcache
 * :234-267).  Allocated rows get count 1.  scratch: ll_kv_alloc_scratch_bytes(n_rows) bytes. */
int64_t ll_kv_alloc_scratch_bytes(int64_t n_rows);
int ll_kv_alloc(int32_t* state, int64_t n_rows, int64_t need, int contiguous_first, int32_t* out_index,
                void* scratch, int64_t* decision, int64_t* free_rows, void* stream);
/* add_ref (delta +1, :302-314) / release_ref (delta -1, :317-333) over ``count`` row ids (int32 or
 * int64; duplicates each count); free_rows is debited for rows leaving 0 and credited for rows
 * reaching 0. */
int ll_kv_ref_update(int32_t* state, int64_t n_rows, const void* index, int64_t count, int idx_width,
                     int delta, int64_t* free_rows, void* stream);

/* ---- one-shot tensor-parallel all-reduce on peer-mapped buffers (a13) -----------------------
 * Replaces the ncclAllReduce of lite_llama/distributed/parallel_state.py:208-213 (in-place SUM of the [tokens, hidden]
 * 16-bit partial sums after every row-parallel projection) for payloads that fit a staging buffer: every rank publishes
 * its tensor in fine-grained memory its peers have mapped, raises a flag in each peer, waits for the peers' flags in its
 * own memory and reads the world - 1 peer buffers over its direct xGMI links, summing in fp32 in RANK ORDER (all ranks
 * get bit-identical results).  One launch, no host call: capturable.  Setup (host, once): ll_tp_shared_alloc the staging
 * buffer [2][stage_elems] 16-bit and ll_tp_oneshot_flag_words(blocks, world) int32 flag words, exchange
 * ll_tp_ipc_export handles through any host channel, ll_tp_ipc_open the peers'.  A peer that does not show up within the
 * spin bound (~10 s) sets bit 0 of the STICKY error word (the LAST flag word) and the launch overwrites its slice with NaN:
 * a lost peer poisons the step instead of producing a plausible partial sum; read the word with ll_tp_error_word. */
int ll_tp_shared_alloc(void** ptr, int64_t bytes);
int ll_tp_shared_free(void* ptr);
int ll_tp_ipc_export(void* ptr, void* handle64);
int ll_tp_ipc_open(const void* handle64, void** ptr);
int ll_tp_ipc_close(void* ptr);
int64_t ll_tp_oneshot_flag_words(int blocks, int world);
/* host read-back of the error word (the last of flag_words); synchronises the device */
int ll_tp_error_word(const void* flags, int64_t flag_words, int32_t* host_out);
/* stage_ptrs / flag_ptrs: HOST arrays of `world` device pointers (entry r = rank r's buffers, mine included);
 * epoch_done: two int32 in this rank's memory, zero before the first launch.  Every rank issues the same call sequence
 * (same count, same blocks).  count <= stage_elems, count % 8 == 0, world <= 8. */
int ll_tp_allreduce_oneshot(void* inout, int64_t count, int dtype, const void* const* stage_ptrs, void* const* flag_ptrs,
                            int rank, int world, int64_t stage_elems, int blocks, int32_t* epoch_done, void* stream);
/* The block's collective fused with its neighbours (models/linear.py:160-161 RowParallelLinear.forward = all_reduce(GEMM),
 * then kernels/skip_rmsnorm.py:192-234): y = skip_rmsnorm(all_reduce_tp(fp16(sum of the S fp32 split-K partials
 * [S][rows][n] that ll_w4a16_matmul_prepacked leaves with epilogue 2)), residual, weight), residual updated in place.
 * Same peer buffers, flag words, epoch counter and `blocks` as ll_tp_allreduce_oneshot (the two may alternate within a
 * step); sums in fp32 in rank order -- bit-identical on all ranks and to GEMM -> ll_tp_allreduce_oneshot ->
 * ll_skip_rmsnorm.  rows * n <= stage_elems, n % 8 == 0, n <= 8192, 1 <= s_count <= 64. */
int ll_tp_allreduce_norm_partials(void* y, const float* partials, int s_count, void* residual, const void* weight,
                                  int64_t rows, int64_t n, float eps, int dtype, const void* const* stage_ptrs,
                                  void* const* flag_ptrs, int rank, int world, int64_t stage_elems, int blocks,
                                  int32_t* epoch_done, void* stream);

/* ---- fp8 KV cache (extension -- the reference's pool is fp16, executor/kv_cache_manager.py:197-216) ----
 * The pool holds OCP e4m3 bytes; stored value * k_scale (v_scale) = K (V) value, static per pool.
 * ll_update_kv_buffer_fp8: a3's scatter with quantisation -- heads [0, k_heads) are K heads (divided by k_scale), the
 * rest V heads; clamp to +-448, round to nearest even; vals fp16 / bf16 [tokens, heads, hd], buf uint8 (strides in
 * bytes), hd % 8 == 0.  ll_flash_decoding_fp8kv: a5 over such a pool -- k_cache / v_cache byte views (strides in bytes),
 * fp16 q / out, d in {64, 128}, counters required (one-launch forms); the fragments are widened to fp16 in registers
 * (exact), k_scale rides on qk_scale and v_scale on the normalisation: the result is ll_flash_decoding's on the
 * widened pool. */
int ll_update_kv_buffer_fp8(const void* vals, const void* select_index, void* buf, int64_t tokens, int heads, int k_heads,
                            int hd, int64_t v_stride_t, int64_t v_stride_h, int64_t b_stride_t, int64_t b_stride_h,
                            float k_scale, float v_scale, int dtype, int idx_width, void* stream);
int ll_flash_decoding_fp8kv(void* out, const void* q, const void* k_cache, const void* v_cache, const int32_t* table,
                            const void* b_req_idx, const void* b_seq_len, float* mid_o, float* mid_lse, int batch, int hq,
                            int hkv, int d, int64_t max_len, float qk_scale, float k_scale, float v_scale,
                            int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                            int64_t v_stride_t, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                            int64_t table_stride_b, int req_width, int seq_width, int32_t* counters, void* stream);

/* ---- block-granular KV paging on the device (extension) -----------------------
 * The reference's pool is token-granular (executor/kv_cache_manager.py:197-216) with the TODO "reshape into
 * [blocks, block_size, ...] to support PagedAttention" (:211); its kernels read the per-token table
 * b_req_tokens_table[req][pos] (executor/model_runner.py:153-218).  These entries hand the SAME pool out in blocks of
 * block_size consecutive rows and derive the per-token table (row = block * block_size + pos % block_size) on the
 * device, so every kernel above runs unchanged.  State, all int32 on the device: free_stack [num_blocks - 1];
 * state[2] = {free blocks, error flags (1 = out of blocks, 2 = a request outgrew its block-table row)};
 * block_table [max_reqs][bt_stride]; req_blocks [max_reqs].  Block num_blocks - 1 is the junk block (rows for padded
 * prefill positions and refused requests); a fresh pool hands out blocks 0, 1, 2, ... in request order.  No host
 * read-back anywhere: all three entries are stream-ordered and may sit inside a captured decode step.  n <= 1024
 * requests per call, each listed once. */
int ll_kv_paged_reset(int32_t* free_stack, int32_t* state, int32_t* req_blocks, int64_t num_blocks, int64_t max_reqs,
                      void* stream);
/* Make request req_idx[i] hold ceil((lens[i] + len_bias) / block_size) blocks (all-or-nothing per call) and write the
 * rows of the grid [n][grid_len]: from_end 0 = positions 0 .. grid_len - 1 (padded prefill grid; positions >= the
 * length get junk rows in select_out and leave the token table alone), from_end 1 = the last grid_len positions
 * (decode append with grid_len 1: position len - 1).  token_table[req][p] = row; select_out [n * grid_len]. */
int ll_kv_paged_extend(int32_t* free_stack, int32_t* state, int32_t* block_table, int64_t bt_stride, int32_t* req_blocks,
                       const int32_t* req_idx, const int32_t* lens, int len_bias, int64_t n, int block_size,
                       int64_t grid_len, int from_end, int32_t* token_table, int64_t tt_stride, int32_t* select_out,
                       int64_t num_blocks, void* stream);
/* Return every block of the listed requests to the stack (request order, block order). */
int ll_kv_paged_release(int32_t* free_stack, int32_t* state, const int32_t* block_table, int64_t bt_stride,
                        int32_t* req_blocks, const int32_t* req_idx, int64_t n, void* stream);

/* ---- load-time ingestion of third-party int4 checkpoint layouts -------------
 * The reference reaches W4A16 only by re-quantising fp16 weights: AutoAWQ / AutoGPTQ tensors map
 * to unknown parameters (models/weights.py:166-173,266-268).  These two entries convert such
 * tensors, on the device and bit-exactly, into the layout a8 consumes: out_qweight int32 [N, K/8]
 * (nibbles sequential along K, LSB first), out_scales / out_zeros fp32 [N, K/group].
 *   AWQ  (AutoAWQ 0.2.x GEMM): qweight int32 [K, N/8] and qzeros int32 [K/g, N/8], eight output
 *        channels per word in the order {0,2,4,6,1,3,5,7}; scales fp16 [K/g, N]; w = (q - z) s.
 *   GPTQ (AutoGPTQ 0.7.x, no act-order): qweight int32 [K/8, N], eight input channels per word,
 *        sequential; qzeros int32 [K/g, N/8] sequential; scales fp16 [K/g, N];
 *        w = (q - (zs + zero_offset)) s with zero_offset 1 for v1 checkpoints, 0 for v2.
 * K % 8 == 0, N % 8 == 0, K % group == 0, else LL_ERR_SHAPE. */
int ll_w4_from_awq(int32_t* out_qweight, float* out_scales, float* out_zeros, const int32_t* qweight,
                   const int32_t* qzeros, const void* scales_f16, int64_t k, int64_t n,
                   int64_t group_size, void* stream);
int ll_w4_from_gptq(int32_t* out_qweight, float* out_scales, float* out_zeros, const int32_t* qweight,
                    const int32_t* qzeros, const void* scales_f16, int64_t k, int64_t n,
                    int64_t group_size, int zero_offset, void* stream);

/* Continuous-batching step metadata in one launch (executor/slot_batch.py:135-169: the steady
 * state of SlotBatch.begin_decode, ``b_seq_len += 1`` then ``cur_select_index = table[b_req_idx,
 * b_seq_len - 1]``).  b_seq_len / b_req_idx are int32 or int64 (idx_width); cur_select_index and
 * the slot table are int32.  Optional (NULL to skip): positions[i] = b_seq_len[i] - 1 (the decode
 * position, continuous_engine.py:404-405) and input_ids[i] = next_tokens[i], all int64. */
int ll_slot_advance(void* b_seq_len, const void* b_req_idx, int32_t* cur_select_index,
                    int64_t* positions, int64_t* input_ids, const int64_t* next_tokens,
                    const int32_t* table, int64_t table_stride_b, int64_t table_stride_s, int batch,
                    int idx_width, void* stream);

/* ---- a16: greedy argmax over logits [rows, n] (engine/sampler.py:227-228) ----- */
int ll_argmax(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t stride_row,
              int dtype, void* stream);
/* Same result through a [rows x chunks] grid + a merge kernel (vocabulary-sized rows at small
 * batch); scratch = rows * chunks * 16 bytes, 8-byte aligned. */
int ll_argmax_split(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t stride_row,
                    int dtype, void* scratch, int chunks, void* stream);

/* ---- sampler row (engine/sampler.py:77-137,199-270) ---------------------------------
 * ll_repetition_penalty: out[b, :] = logits[b, :] with, for every generated token (token_ids[b, j]
 * where mask[b, j] != 0), logit / penalty if >= 0 else logit * penalty, computed from the ORIGINAL
 * logit (repeats count once).  fp32 arithmetic, one rounding to out_dtype (F16->F16, F16->F32,
 * BF16->BF16, BF16->F32, F32->F32: torch's promotion for a scalar / float32-tensor penalty).
 * penalty_rows [batch] or NULL (then penalty_scalar).  out must not alias logits.  mask: 1 byte / entry.
 * ll_sample_top_p: per row softmax(logits / temperature) -> nucleus (tokens whose strictly more
 * probable mass <= top_p) -> one draw from it by inverse CDF in index order at uniform[b] in [0, 1);
 * rows with greedy[b] != 0 (optional) return the first argmax.  out int64 [batch]. */
int ll_repetition_penalty(void* out, const void* logits, const int64_t* token_ids, const void* mask,
                          const float* penalty_rows, float penalty_scalar, int64_t batch, int64_t vocab,
                          int64_t span, int64_t logits_stride, int64_t out_stride, int64_t ids_stride,
                          int64_t mask_stride, int in_dtype, int out_dtype, void* stream);
int ll_sample_top_p(int64_t* out, const void* logits, const float* temperature, const float* top_p,
                    const float* uniform, const void* greedy, int64_t batch, int64_t vocab,
                    int64_t stride, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LITE_LLAMA_AMD_H */
