// Decode-step fusions around smoothquant_matmul (a10) -- reference lite_llama/kernels/quantization/w8a8.py:28-217 and the
// callers in models/quantization/methods/w8a8.py:16-38.  The reference runs, per projection, a per-token quantiser launch,
// the int8 GEMM and its scale epilogue; between two projections sit skip_rmsnorm / swiglu launches of their own.  At decode
// shapes every one of those is a latency-bound launch of ~5 - 8 us next to 10 - 30 us GEMMs.  Here:
//   * ll_skip_rmsnorm_q8: (int32 split-K planes of the previous W8A8 projection -> exact int32 sum -> * a_scale[m] * w_scale[n]
//     (+ bias) -> fp16: the value smoothquant_matmul returns) -> skip_rmsnorm -> per-token int8 quantiser of the NEXT
//     projection -- one launch, one row per workgroup, the row never leaves registers;
//   * ll_w8a8_finish_swiglu: the fused gate|up projection's planes -> scale epilogue -> silu(gate) * up, element-wise.
// Same arithmetic, in the same order, as the separate launches (dense8_finish, skip_rmsnorm_cached, quant_act_kernel,
// swiglu_kernel): the tests compare bit for bit.
#include "common.h"

namespace {

// A product that is rounded to fp32 BEFORE it is narrowed to fp16 (what the separate launches -- and the reference's
// `(x * r).to(fp16)` -- compute).  Without the opaque move hipcc may select v_fma_mixlo_f16 for fptrunc(fmul): ONE rounding,
// which differs from the two-step value once in ~2^13 elements (measured: 8 of 131072 outputs of this file's norm differed
// from skip_rmsnorm_cached's until the product was pinned).
__device__ __forceinline__ float q8_f32(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int TPR>
__device__ __forceinline__ float q8_row_sum(float v, float* lds) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[wid] = v;
  __syncthreads();
  return lds[0] + lds[1] + lds[2] + lds[3];
}

// SMAX == 0: the input is a finished fp16 tensor x [rows][n];  SMAX > 0: int32 planes [S][rows][n] + scales (+ bias).
// HAS_RES as skip_rmsnorm (residual updated in place).  y (fp16) and q / q_scale (int8 rows + per-token scale) are each optional.
template <int VPT, int SMAX, bool HAS_RES>
__global__ __launch_bounds__(256) void skip_rmsnorm_q8_kernel(uint16_t* __restrict__ y, int8_t* __restrict__ q,
                                                              float* __restrict__ q_scale, const uint16_t* __restrict__ x,
                                                              const int32_t* __restrict__ part, int s_count,
                                                              const float* __restrict__ a_scale, const float* __restrict__ w_scale,
                                                              const uint16_t* __restrict__ bias, uint16_t* __restrict__ r,
                                                              const uint16_t* __restrict__ w, int64_t rows, int n, float eps) {
  constexpr int DT = LL_F16;
  constexpr int SL = SMAX > 0 ? SMAX : 1;
  __shared__ float lds[4];
  __shared__ float lds_max[4];
  const int tr = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float nf = (float)n;
  const int64_t plane = rows * (int64_t)n;
  // every load of the launch is issued here, unconditionally (skip_rmsnorm_cached's rule)
  U16x8 rv[VPT], wv[VPT], xv[VPT], bv[VPT];
  i32x4 pv[VPT][SL][2];
  f32x4 ws[VPT][2];
  bool ok[VPT];
  const float as = SMAX > 0 ? a_scale[row] : 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int col = (v * 256 + tr) * 8;
    ok[v] = col < n;
    const int64_t off = ok[v] ? row * n + col : 0;
    const int c0 = ok[v] ? col : 0;
    if constexpr (HAS_RES) rv[v] = *reinterpret_cast<const U16x8*>(r + off);
    wv[v] = *reinterpret_cast<const U16x8*>(w + c0);
    if constexpr (SMAX > 0) {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        const int32_t* src = part + (s < s_count ? s : 0) * plane + off;  // slots >= s_count re-read plane 0 and are dropped below
        pv[v][s][0] = *reinterpret_cast<const i32x4*>(src);
        pv[v][s][1] = *reinterpret_cast<const i32x4*>(src + 4);
      }
      ws[v][0] = *reinterpret_cast<const f32x4*>(w_scale + c0);
      ws[v][1] = *reinterpret_cast<const f32x4*>(w_scale + c0 + 4);
      bv[v] = *reinterpret_cast<const U16x8*>(bias ? bias + c0 : w + c0);  // (a pointer select: no load inside a branch)
    } else {
      xv[v] = *reinterpret_cast<const U16x8*>(x + off);
    }
  }
  float sv[VPT][8];
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float xin;
      if constexpr (SMAX > 0) {
        int a = 0;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) a += s < s_count ? pv[v][s][i >> 2][i & 3] : 0;
        const float f = q8_f32(((float)a * as) * ws[v][i >> 2][i & 3]);  // (acc.f32 * a_scale[m]) * w_scale[n], w8a8.py:118-120
        xin = to_f32<DT>(from_f32<DT>(q8_f32(bias ? f + to_f32<DT>(bv[v].v[i]) : f)));
      } else {
        xin = to_f32<DT>(xv[v].v[i]);
      }
      float s = xin;
      if constexpr (HAS_RES) {
        s += to_f32<DT>(rv[v].v[i]);
        rv[v].v[i] = from_f32<DT>(q8_f32(s));
      }
      sv[v][i] = ok[v] ? s : 0.f;
    }
    if constexpr (HAS_RES) {
      if (ok[v]) *reinterpret_cast<U16x8*>(r + row * n + (v * 256 + tr) * 8) = rv[v];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) ssq += sv[v][i] * sv[v][i] / nf;
  }
  const float var = q8_row_sum<256>(ssq, lds);
  const float rrms = 1.0f / sqrtf(var + eps);
  U16x8 yv[VPT];
  float amax = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      yv[v].v[i] = mul_storage<DT>(from_f32<DT>(q8_f32(sv[v][i] * rrms)), wv[v].v[i]);
      if (ok[v]) amax = fmaxf(amax, fabsf(to_f32<DT>(yv[v].v[i])));
    }
    if (y && ok[v]) *reinterpret_cast<U16x8*>(y + row * n + (v * 256 + tr) * 8) = yv[v];
  }
  if (!q) return;  // (kernel argument: uniform)
  // per-token quantiser of the consumer (w8a8.py:34-68): scale = absmax / 127 (1 if 0), q = trunc(y / scale)
  amax = wave_max(amax);
  if ((tr & 63) == 0) lds_max[tr >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(lds_max[0], lds_max[1]), fmaxf(lds_max[2], lds_max[3]));
  float scale = amax / 127.0f;
  scale = scale > 0.f ? scale : 1.0f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (!ok[v]) continue;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int qi = (int)truncf(to_f32<DT>(yv[v].v[i]) / scale);
      const uint32_t b = (uint32_t)(uint8_t)(int8_t)qi;
      if (i < 4) lo |= b << (8 * i);
      else hi |= b << (8 * (i - 4));
    }
    *reinterpret_cast<uint2*>(q + row * n + (v * 256 + tr) * 8) = uint2{lo, hi};
  }
  if (tr == 0) q_scale[row] = scale;
}

// out[m][j] = silu(g) * u with (g, u) = finished values of the planes' columns (2 j, 2 j + 1) -- the fused gate|up projection
// keeps its rows interleaved (linear.py::MergedColumnLinear).  One thread per 4 plane columns = 2 outputs.
__global__ __launch_bounds__(256) void w8a8_finish_swiglu_kernel(uint16_t* __restrict__ out, const int32_t* __restrict__ part,
                                                                 int splits, int64_t m, int64_t n,
                                                                 const float* __restrict__ a_scale,
                                                                 const float* __restrict__ w_scale) {
  constexpr int DT = LL_F16;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= m * n) return;
  const int64_t row = i / n, col = i - row * n;
  i32x4 a = {0, 0, 0, 0};
  for (int s0 = 0; s0 < splits; s0 += 8) {  // eight planes per round, requested together; summed in split order (exact anyway)
    i32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const i32x4*>(part + (int64_t)(s0 + u < splits ? s0 + u : s0) * m * n + i);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < splits) a += t[u];
  }
  const float as = a_scale[row];
  const f32x4 wsc = *reinterpret_cast<const f32x4*>(w_scale + col);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = to_f32<DT>(from_f32<DT>(q8_f32(((float)a[e] * as) * wsc[e])));  // the fp16 the projection stores
  uint16_t o[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float g = v[2 * e];
    o[e] = from_f32<DT>(q8_f32(g * ll_sigmoidf(g) * v[2 * e + 1]));  // swiglu_kernel's arithmetic
  }
  *reinterpret_cast<uint32_t*>(out + row * (n / 2) + col / 2) = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
}

// per-token quantiser, one row per workgroup, the row cached in registers (k <= 256 * 8 * VPT)
template <int VPT>
__global__ __launch_bounds__(256) void quant_act_cached_kernel(int8_t* __restrict__ q, float* __restrict__ a_scale,
                                                               const uint16_t* __restrict__ x, int64_t k, int64_t x_stride) {
  __shared__ float red[4];
  const int tr = threadIdx.x;
  const int64_t row = blockIdx.x;
  U16x8 xv[VPT];
  bool ok[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int64_t col = (int64_t)(v * 256 + tr) * 8;
    ok[v] = col < k;
    xv[v] = *reinterpret_cast<const U16x8*>(x + row * x_stride + (ok[v] ? col : 0));
  }
  float amax = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (ok[v]) amax = fmaxf(amax, fabsf(f16_bits_to_f32(xv[v].v[i])));
  amax = wave_max(amax);
  if ((tr & 63) == 0) red[tr >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float scale = amax / 127.0f;
  scale = scale > 0.f ? scale : 1.0f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (!ok[v]) continue;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int qi = (int)truncf(f16_bits_to_f32(xv[v].v[i]) / scale);
      const uint32_t b = (uint32_t)(uint8_t)(int8_t)qi;
      if (i < 4) lo |= b << (8 * i);
      else hi |= b << (8 * (i - 4));
    }
    *reinterpret_cast<uint2*>(q + row * k + (int64_t)(v * 256 + tr) * 8) = uint2{lo, hi};
  }
  if (tr == 0) a_scale[row] = scale;
}

}  // namespace

// The vectorised per-token quantiser (gemm_wq.hip::ll_quantize_activations_int8 dispatches here): 1 when launched, 0 when the
// shape / alignment is not served (k % 8, 16-byte rows, k <= 16384).
extern "C" int ll_quant_act_cached_try(int8_t* q, float* a_scale, const void* x, int64_t m, int64_t k, int64_t x_stride_m,
                                       void* stream) {
  if (k % 8 != 0 || k > 16384 || x_stride_m % 8 != 0 || !ll_aligned16(x) || ((uintptr_t)q & 7) != 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)m);
  auto xx = (const uint16_t*)x;
  if (k <= 2048) quant_act_cached_kernel<1><<<grid, 256, 0, st>>>(q, a_scale, xx, k, x_stride_m);
  else if (k <= 4096) quant_act_cached_kernel<2><<<grid, 256, 0, st>>>(q, a_scale, xx, k, x_stride_m);
  else if (k <= 8192) quant_act_cached_kernel<4><<<grid, 256, 0, st>>>(q, a_scale, xx, k, x_stride_m);
  else quant_act_cached_kernel<8><<<grid, 256, 0, st>>>(q, a_scale, xx, k, x_stride_m);
  return hipGetLastError() == hipSuccess ? 1 : LL_ERR_LAUNCH;
}

// skip_rmsnorm (fp16) for a W8A8 block.  Input: EITHER x [rows][n] fp16 (planes NULL) OR the int32 split-K planes
// [s_count][rows][n] of a W8A8 projection (ll_w8a8_partials) with its per-token / per-channel scales and optional bias
// (x NULL): then the normalised input is fp16(((float)sum * a_scale[m]) * w_scale[n] (+ bias)), the value
// ll_w8a8_matmul stores.  residual [rows][n] (updated in place) or NULL.  Outputs, each optional: y [rows][n] fp16, and
// q [rows][n] int8 + q_scale [rows] = ll_quantize_activations_int8(y).  n % 8 == 0, n <= 8192, s_count <= 12.
extern "C" int ll_skip_rmsnorm_q8(void* y, int8_t* q, float* q_scale, const void* x, const int32_t* planes, int s_count,
                                  const float* a_scale, const float* w_scale, const void* bias, void* residual,
                                  const void* weight, int64_t rows, int64_t n, float eps, void* stream) {
  if (rows < 0 || n <= 0 || n % 8 != 0 || n > 8192) return LL_ERR_SHAPE;
  if ((x != nullptr) == (planes != nullptr) || !weight || (!y && !q) || (q && !q_scale)) return LL_ERR_ARG;
  if (planes && (s_count < 1 || s_count > 12 || !a_scale || !w_scale)) return LL_ERR_ARG;
  if (!ll_aligned16(y) || !ll_aligned16(x) || !ll_aligned16(planes) || !ll_aligned16(residual) || !ll_aligned16(weight) ||
      !ll_aligned16(w_scale) || !ll_aligned16(bias) || ((uintptr_t)q & 7) != 0)
    return LL_ERR_ARG;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nv = (int)(n / 8);
#define LL_Q8_CASE(VPT, SMAX, RES)                                                                                         \
  skip_rmsnorm_q8_kernel<VPT, SMAX, RES><<<dim3((unsigned)rows), 256, 0, st>>>(                                            \
      (uint16_t*)y, q, q_scale, (const uint16_t*)x, planes, s_count, a_scale, w_scale, (const uint16_t*)bias,              \
      (uint16_t*)residual, (const uint16_t*)weight, rows, (int)n, eps)
#define LL_Q8_RES(VPT, SMAX) \
  if (residual) LL_Q8_CASE(VPT, SMAX, true); else LL_Q8_CASE(VPT, SMAX, false)
#define LL_Q8_S(VPT)                                                                        \
  if (!planes) { LL_Q8_RES(VPT, 0); }                                                       \
  else if (s_count <= 4) { LL_Q8_RES(VPT, 4); } else if (s_count <= 8) { LL_Q8_RES(VPT, 8); } \
  else { LL_Q8_RES(VPT, 12); }
  if (nv <= 256) { LL_Q8_S(1) } else if (nv <= 512) { LL_Q8_S(2) } else { LL_Q8_S(4) }
#undef LL_Q8_S
#undef LL_Q8_RES
#undef LL_Q8_CASE
  return LL_LAUNCH_CHECK();
}

// out [m][n / 2] fp16 = silu(gate) * up over the int32 planes [s_count][m][n] of a fused gate|up W8A8 projection whose rows
// are interleaved (gate_j, up_j): gate / up = fp16(((float)sum * a_scale[m]) * w_scale[col]) as ll_w8a8_matmul stores them.
extern "C" int ll_w8a8_finish_swiglu(void* out, const int32_t* planes, int s_count, const float* a_scale, const float* w_scale,
                                     int64_t m, int64_t n, void* stream) {
  if (m < 0 || n <= 0 || n % 4 != 0 || s_count < 1) return LL_ERR_SHAPE;
  if (!out || !planes || !a_scale || !w_scale || !ll_aligned16(planes) || !ll_aligned16(w_scale) || ((uintptr_t)out & 3) != 0)
    return LL_ERR_ARG;
  if (m == 0) return LL_OK;
  const int64_t quads = m * n / 4;
  w8a8_finish_swiglu_kernel<<<dim3((unsigned)((quads + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (uint16_t*)out, planes, s_count, m, n, a_scale, w_scale);
  return LL_LAUNCH_CHECK();
}
