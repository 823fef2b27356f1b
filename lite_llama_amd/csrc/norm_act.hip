// Bandwidth-bound row kernels of the decode step: skip_rmsnorm (a1), swiglu (a7),
// rope (a2), silu_and_mul / moe_sum (a11 pieces), greedy argmax (a16).
// All HBM-bound: 16-byte vector loads, wave64 shuffles, one pass over the data.
#include "common.h"

// --------------------------------------------------------------------------- //
// skip_rmsnorm  -- reference lite_llama/kernels/skip_rmsnorm.py:126-234
// --------------------------------------------------------------------------- //
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* lds) {
  if constexpr (TPR <= 64) {
#pragma unroll
    for (int off = TPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  } else {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wid] = v;
    __syncthreads();
    float t = lds[0] + lds[1] + lds[2] + lds[3];
    return t;
  }
}

// Cached variant: the whole row lives in registers (VPT vectors of 8 per thread).
template <int DT, int TPR, int VPT, bool HAS_RES>
__global__ __launch_bounds__(256) void skip_rmsnorm_cached(uint16_t* __restrict__ y,
                                                           const uint16_t* __restrict__ x,
                                                           uint16_t* __restrict__ r,
                                                           const uint16_t* __restrict__ w,
                                                           int64_t rows, int n, float eps) {
  __shared__ float lds[4];
  constexpr int RPB = 256 / TPR;
  const int tr = threadIdx.x % TPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
  const bool active = row < rows;
  const float nf = (float)n;
  // Every load of the kernel is issued here, unconditionally (a lane without work re-reads element
  // 0): loads inside branches make hipcc drain the memory counter per branch, which turns the
  // kernel into a chain of dependent round trips (x/residual per vector, then the weight per
  // vector after the reduction) -- at decode sizes the kernel is nothing but latency.
  U16x8 xv[VPT], rv[VPT], wv[VPT];
  bool ok[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int col = (v * TPR + tr) * 8;
    ok[v] = active && col < n;
    const int64_t off = ok[v] ? row * n + col : 0;
    xv[v] = *reinterpret_cast<const U16x8*>(x + off);
    if constexpr (HAS_RES) rv[v] = *reinterpret_cast<const U16x8*>(r + off);
    wv[v] = *reinterpret_cast<const U16x8*>(w + (ok[v] ? col : 0));
  }
  float s[VPT][8];
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float sv = to_f32<DT>(xv[v].v[i]);
      if constexpr (HAS_RES) {
        sv += to_f32<DT>(rv[v].v[i]);
        rv[v].v[i] = from_f32<DT>(sv);
      }
      s[v][i] = ok[v] ? sv : 0.f;
    }
    if constexpr (HAS_RES) {
      if (ok[v]) *reinterpret_cast<U16x8*>(r + row * n + (v * TPR + tr) * 8) = rv[v];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) ssq += s[v][i] * s[v][i] / nf;
  }
  const float var = row_sum<TPR>(ssq, lds);
  const float rrms = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (ok[v]) {
      U16x8 yv;
#pragma unroll
      for (int i = 0; i < 8; ++i) yv.v[i] = mul_storage<DT>(from_f32<DT>(s[v][i] * rrms), wv[v].v[i]);
      *reinterpret_cast<U16x8*>(y + row * n + (v * TPR + tr) * 8) = yv;
    }
  }
}

// Generic two-pass variant (any n, any alignment when VEC == 1): one row per block.
template <int DT, int VEC, bool HAS_RES>
__global__ __launch_bounds__(256) void skip_rmsnorm_twopass(uint16_t* __restrict__ y,
                                                            const uint16_t* __restrict__ x,
                                                            uint16_t* __restrict__ r,
                                                            const uint16_t* __restrict__ w,
                                                            int64_t rows, int n, float eps) {
  __shared__ float lds[4];
  const int64_t row = blockIdx.x;
  const float nf = (float)n;
  const int nvec = n / VEC;
  float ssq = 0.f;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    uint16_t xv[VEC], rv[VEC];
    VecIO<VEC>::load(x + row * n + (int64_t)v * VEC, xv);
    if constexpr (HAS_RES) VecIO<VEC>::load(r + row * n + (int64_t)v * VEC, rv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float sv = to_f32<DT>(xv[i]);
      if constexpr (HAS_RES) sv += to_f32<DT>(rv[i]);
      ssq += sv * sv / nf;
    }
  }
  const float var = row_sum<256>(ssq, lds);
  const float rrms = 1.0f / sqrtf(var + eps);
  for (int v = threadIdx.x; v < nvec; v += 256) {
    uint16_t xv[VEC], rv[VEC], wv[VEC], yv[VEC];
    VecIO<VEC>::load(x + row * n + (int64_t)v * VEC, xv);
    if constexpr (HAS_RES) VecIO<VEC>::load(r + row * n + (int64_t)v * VEC, rv);
    VecIO<VEC>::load(w + (int64_t)v * VEC, wv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float sv = to_f32<DT>(xv[i]);
      if constexpr (HAS_RES) {
        sv += to_f32<DT>(rv[i]);
        rv[i] = from_f32<DT>(sv);
      }
      yv[i] = mul_storage<DT>(from_f32<DT>(sv * rrms), wv[i]);
    }
    if constexpr (HAS_RES) VecIO<VEC>::store(r + row * n + (int64_t)v * VEC, rv);
    VecIO<VEC>::store(y + row * n + (int64_t)v * VEC, yv);
  }
}

template <int DT, bool HAS_RES>
static int launch_skip_rmsnorm(uint16_t* y, const uint16_t* x, uint16_t* r, const uint16_t* w,
                               int64_t rows, int n, float eps, hipStream_t st) {
  const bool vec = (n % 8 == 0) && ll_aligned16(y) && ll_aligned16(x) && ll_aligned16(w) &&
                   (!HAS_RES || ll_aligned16(r));
  if (!vec) {
    skip_rmsnorm_twopass<DT, 1, HAS_RES><<<dim3((unsigned)rows), 256, 0, st>>>(y, x, r, w, rows, n, eps);
    return LL_LAUNCH_CHECK();
  }
  const int nv = n / 8;
#define LL_NORM_CASE(TPR, VPT)                                                              \
  skip_rmsnorm_cached<DT, TPR, VPT, HAS_RES>                                                \
      <<<dim3((unsigned)((rows + (256 / TPR) - 1) / (256 / TPR))), 256, 0, st>>>(y, x, r, w, rows, n, eps)
  if (nv <= 16) LL_NORM_CASE(16, 1);
  else if (nv <= 32) LL_NORM_CASE(32, 1);
  else if (nv <= 64) LL_NORM_CASE(64, 1);
  else if (nv <= 256) LL_NORM_CASE(256, 1);
  else if (nv <= 512) LL_NORM_CASE(256, 2);
  else if (nv <= 1024) LL_NORM_CASE(256, 4);
  else skip_rmsnorm_twopass<DT, 8, HAS_RES><<<dim3((unsigned)rows), 256, 0, st>>>(y, x, r, w, rows, n, eps);
#undef LL_NORM_CASE
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_skip_rmsnorm(void* y, const void* x, void* residual, const void* weight,
                               int64_t rows, int64_t n, float eps, int dtype, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (rows < 0 || n <= 0 || n > 65536) return LL_ERR_SHAPE;  // utils.py:48-54 (MAX_FUSED_SIZE)
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  auto yy = (uint16_t*)y;
  auto xx = (const uint16_t*)x;
  auto rr = (uint16_t*)residual;
  auto ww = (const uint16_t*)weight;
  if (dtype == LL_F16)
    return residual ? launch_skip_rmsnorm<LL_F16, true>(yy, xx, rr, ww, rows, (int)n, eps, st)
                    : launch_skip_rmsnorm<LL_F16, false>(yy, xx, rr, ww, rows, (int)n, eps, st);
  return residual ? launch_skip_rmsnorm<LL_BF16, true>(yy, xx, rr, ww, rows, (int)n, eps, st)
                  : launch_skip_rmsnorm<LL_BF16, false>(yy, xx, rr, ww, rows, (int)n, eps, st);
}

// skip_rmsnorm whose input projection was left as S fp32 split-K partials [S][rows][n] (gemm_w4_v3.hip, epilogue 2):
// x = fp16(sum_s P[s]) -- exactly the value the projection's own epilogue would have rounded and stored --
// then the arithmetic of skip_rmsnorm_cached.  One row per block; every load issued up front.
template <int DT, int VPT, int SMAX>
__global__ __launch_bounds__(256) void skip_rmsnorm_partials_kernel(uint16_t* __restrict__ y, const float* __restrict__ part,
                                                                    int s_count, uint16_t* __restrict__ r,
                                                                    const uint16_t* __restrict__ w, int64_t rows, int n,
                                                                    float eps) {
  __shared__ float lds[4];
  const int tr = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float nf = (float)n;
  const int64_t plane = rows * (int64_t)n;
  U16x8 rv[VPT], wv[VPT];
  f32x4 pv[VPT][SMAX][2];
  bool ok[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int col = (v * 256 + tr) * 8;
    ok[v] = col < n;
    const int64_t off = ok[v] ? row * n + col : 0;
    rv[v] = *reinterpret_cast<const U16x8*>(r + off);
    wv[v] = *reinterpret_cast<const U16x8*>(w + (ok[v] ? col : 0));
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const float* src = part + (s < s_count ? s : 0) * plane + off;  // slots >= s_count re-read slot 0 and are dropped below
      pv[v][s][0] = *reinterpret_cast<const f32x4*>(src);
      pv[v][s][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  }
  float sv[VPT][8];
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) a += s < s_count ? pv[v][s][i >> 2][i & 3] : 0.f;
      float x = to_f32<DT>(from_f32<DT>(a)) + to_f32<DT>(rv[v].v[i]);
      rv[v].v[i] = from_f32<DT>(x);
      sv[v][i] = ok[v] ? x : 0.f;
    }
    if (ok[v]) *reinterpret_cast<U16x8*>(r + row * n + (v * 256 + tr) * 8) = rv[v];
#pragma unroll
    for (int i = 0; i < 8; ++i) ssq += sv[v][i] * sv[v][i] / nf;
  }
  const float var = row_sum<256>(ssq, lds);
  const float rrms = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (ok[v]) {
      U16x8 yv;
#pragma unroll
      for (int i = 0; i < 8; ++i) yv.v[i] = mul_storage<DT>(from_f32<DT>(sv[v][i] * rrms), wv[v].v[i]);
      *reinterpret_cast<U16x8*>(y + row * n + (v * 256 + tr) * 8) = yv;
    }
  }
}

extern "C" int ll_skip_rmsnorm_partials(void* y, const float* partials, int s_count, void* residual, const void* weight,
                                        int64_t rows, int64_t n, float eps, int dtype, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (rows < 0 || n <= 0 || n % 8 != 0 || n > 8192 || s_count < 1 || s_count > 12) return LL_ERR_SHAPE;
  if (!y || !partials || !residual || !weight || !ll_aligned16(y) || !ll_aligned16(partials) || !ll_aligned16(residual) ||
      !ll_aligned16(weight))
    return LL_ERR_ARG;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nv = (int)(n / 8);
#define LL_PART_CASE(DT, VPT, SMAX)                                                                             \
  skip_rmsnorm_partials_kernel<DT, VPT, SMAX><<<dim3((unsigned)rows), 256, 0, st>>>(                            \
      (uint16_t*)y, partials, s_count, (uint16_t*)residual, (const uint16_t*)weight, rows, (int)n, eps)
// (SMAX = the number of plane loads issued per column group: slots >= s_count re-read plane 0 -- an exact-fit instance for the
// 8-plane split of the XCD-aware slice map saves a third of the launch's load requests)
#define LL_PART_S(DT, VPT)                                                   \
  if (s_count <= 4) LL_PART_CASE(DT, VPT, 4); else if (s_count <= 6) LL_PART_CASE(DT, VPT, 6); \
  else if (s_count <= 8) LL_PART_CASE(DT, VPT, 8); else LL_PART_CASE(DT, VPT, 12)
#define LL_PART_DT(DT)                                                       \
  if (nv <= 256) { LL_PART_S(DT, 1); } else if (nv <= 512) { LL_PART_S(DT, 2); } else { LL_PART_S(DT, 4); }
  if (dtype == LL_F16) { LL_PART_DT(LL_F16) } else { LL_PART_DT(LL_BF16) }
#undef LL_PART_DT
#undef LL_PART_S
#undef LL_PART_CASE
  return LL_LAUNCH_CHECK();
}

// skip_rmsnorm whose input is the fused-MoE block's per-slot output [rows][k][n] (16-bit: the down projection's rows with the
// router weight folded in): x = moe_sum's value -- fp32 sum over the k slots in order, one rounding (fused_moe.py:318-335) --
// then the arithmetic of skip_rmsnorm_cached.  Replaces the moe_sum launch + the norm launch.  One row per block.
template <int DT, int VPT, int KMAX>
__global__ __launch_bounds__(256) void skip_rmsnorm_slots_kernel(uint16_t* __restrict__ y, const uint16_t* __restrict__ slots,
                                                                 int k_count, uint16_t* __restrict__ r,
                                                                 const uint16_t* __restrict__ w, int64_t rows, int n, float eps) {
  __shared__ float lds[4];
  const int tr = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float nf = (float)n;
  U16x8 rv[VPT], wv[VPT], hv[VPT][KMAX];
  bool ok[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    const int col = (v * 256 + tr) * 8;
    ok[v] = col < n;
    const int c0 = ok[v] ? col : 0;
    rv[v] = *reinterpret_cast<const U16x8*>(r + row * n + c0);
    wv[v] = *reinterpret_cast<const U16x8*>(w + c0);
#pragma unroll
    for (int s = 0; s < KMAX; ++s)  // slots >= k_count re-read slot 0 and are dropped below
      hv[v][s] = *reinterpret_cast<const U16x8*>(slots + (row * k_count + (s < k_count ? s : 0)) * (int64_t)n + c0);
  }
  float sv[VPT][8];
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < KMAX; ++s) a += s < k_count ? to_f32<DT>(hv[v][s].v[i]) : 0.f;
      float x = to_f32<DT>(from_f32<DT>(a)) + to_f32<DT>(rv[v].v[i]);
      rv[v].v[i] = from_f32<DT>(x);
      sv[v][i] = ok[v] ? x : 0.f;
    }
    if (ok[v]) *reinterpret_cast<U16x8*>(r + row * n + (v * 256 + tr) * 8) = rv[v];
#pragma unroll
    for (int i = 0; i < 8; ++i) ssq += sv[v][i] * sv[v][i] / nf;
  }
  const float var = row_sum<256>(ssq, lds);
  const float rrms = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
    if (ok[v]) {
      U16x8 yv;
#pragma unroll
      for (int i = 0; i < 8; ++i) yv.v[i] = mul_storage<DT>(from_f32<DT>(sv[v][i] * rrms), wv[v].v[i]);
      *reinterpret_cast<U16x8*>(y + row * n + (v * 256 + tr) * 8) = yv;
    }
  }
}

extern "C" int ll_skip_rmsnorm_slots(void* y, const void* slots, int k_count, void* residual, const void* weight, int64_t rows,
                                     int64_t n, float eps, int dtype, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (rows < 0 || n <= 0 || n % 8 != 0 || n > 8192 || k_count < 1 || k_count > 8) return LL_ERR_SHAPE;
  if (!y || !slots || !residual || !weight || !ll_aligned16(y) || !ll_aligned16(slots) || !ll_aligned16(residual) ||
      !ll_aligned16(weight))
    return LL_ERR_ARG;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nv = (int)(n / 8);
#define LL_SLOT_CASE(DT, VPT, KMAX)                                                                          \
  skip_rmsnorm_slots_kernel<DT, VPT, KMAX><<<dim3((unsigned)rows), 256, 0, st>>>(                            \
      (uint16_t*)y, (const uint16_t*)slots, k_count, (uint16_t*)residual, (const uint16_t*)weight, rows, (int)n, eps)
#define LL_SLOT_K(DT, VPT) \
  if (k_count <= 2) LL_SLOT_CASE(DT, VPT, 2); else if (k_count <= 4) LL_SLOT_CASE(DT, VPT, 4); else LL_SLOT_CASE(DT, VPT, 8)
#define LL_SLOT_DT(DT) \
  if (nv <= 256) { LL_SLOT_K(DT, 1); } else if (nv <= 512) { LL_SLOT_K(DT, 2); } else { LL_SLOT_K(DT, 4); }
  if (dtype == LL_F16) { LL_SLOT_DT(LL_F16) } else { LL_SLOT_DT(LL_BF16) }
#undef LL_SLOT_DT
#undef LL_SLOT_K
#undef LL_SLOT_CASE
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// swiglu_forward -- reference lite_llama/kernels/swiglu.py:24-65
// silu_and_mul   -- reference lite_llama/kernels/fused_moe.py:298-315
// --------------------------------------------------------------------------- //

// MODE 0: c[row, col] = silu(a[row, col]) * b[row, col]
// MODE 1: c[row, col] = silu(x[row, col]) * x[row, n + col]   (x has 2n columns)
// MODE 2: c[row, col] = silu(x[row, 2 col]) * x[row, 2 col + 1]  (x has 2n columns: the output of a fused gate|up projection
//         whose rows were interleaved (gate_j, up_j) at load time, called with more rows than the fused launch serves)
template <int DT, int VEC, int MODE>
__global__ __launch_bounds__(256) void swiglu_kernel(uint16_t* __restrict__ c,
                                                     const uint16_t* __restrict__ a,
                                                     const uint16_t* __restrict__ b, int64_t rows,
                                                     int64_t n) {
  const int64_t nvec = n / VEC;
  const int64_t total = rows * nvec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / nvec, col = (i % nvec) * VEC;
    uint16_t av[VEC], bv[VEC], cv[VEC];
    if constexpr (MODE == 0) {
      VecIO<VEC>::load(a + row * n + col, av);
      VecIO<VEC>::load(b + row * n + col, bv);
    } else if constexpr (MODE == 1) {
      VecIO<VEC>::load(a + row * 2 * n + col, av);
      VecIO<VEC>::load(a + row * 2 * n + n + col, bv);
    } else {
      uint16_t lo[VEC], hi[VEC];  // 2 VEC consecutive values = VEC (gate, up) pairs
      VecIO<VEC>::load(a + row * 2 * n + 2 * col, lo);
      if constexpr (VEC > 1) {
        VecIO<VEC>::load(a + row * 2 * n + 2 * col + VEC, hi);
#pragma unroll
        for (int j = 0; j < VEC / 2; ++j) {
          av[j] = lo[2 * j]; bv[j] = lo[2 * j + 1];
          av[VEC / 2 + j] = hi[2 * j]; bv[VEC / 2 + j] = hi[2 * j + 1];
        }
      } else {
        av[0] = lo[0];
        bv[0] = a[row * 2 * n + 2 * col + 1];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float g = to_f32<DT>(av[j]);
      cv[j] = from_f32<DT>(ll_silu_mul_f32(g, to_f32<DT>(bv[j])));
    }
    VecIO<VEC>::store(c + row * n + col, cv);
  }
}

template <int MODE>
static int launch_swiglu(void* c, const void* a, const void* b, int64_t rows, int64_t n, int dtype,
                         void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (rows < 0 || n <= 0) return LL_ERR_SHAPE;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (n % 8 == 0) && ll_aligned16(c) && ll_aligned16(a) && (MODE != 0 || ll_aligned16(b));
  const int64_t total = rows * (vec ? n / 8 : n);
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  auto cc = (uint16_t*)c;
  auto aa = (const uint16_t*)a;
  auto bb = (const uint16_t*)b;
#define LL_SW(DT, VEC) swiglu_kernel<DT, VEC, MODE><<<dim3(grid), 256, 0, st>>>(cc, aa, bb, rows, n)
  if (dtype == LL_F16) {
    if (vec) LL_SW(LL_F16, 8); else LL_SW(LL_F16, 1);
  } else {
    if (vec) LL_SW(LL_BF16, 8); else LL_SW(LL_BF16, 1);
  }
#undef LL_SW
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_swiglu(void* c, const void* a, const void* b, int64_t rows, int64_t n, int dtype,
                         void* stream) {
  return launch_swiglu<0>(c, a, b, rows, n, dtype, stream);
}
extern "C" int ll_silu_and_mul(void* out, const void* x, int64_t rows, int64_t n, int dtype,
                               void* stream) {
  return launch_swiglu<1>(out, x, nullptr, rows, n, dtype, stream);
}
extern "C" int ll_silu_and_mul_pairs(void* out, const void* x, int64_t rows, int64_t n, int dtype, void* stream) {
  return launch_swiglu<2>(out, x, nullptr, rows, n, dtype, stream);
}

// --------------------------------------------------------------------------- //
// relu / leaky_relu / tanh / gelu -- the element-wise names of lite_llama/kernels/activations.py:19-57 (device helpers of
// the reference's Triton kernels; no caller in its models).  kind 0 relu: max(0, x); 1 leaky_relu: x >= 0 ? x : 0.01 x
// (the slope in the storage dtype, as the reference casts it); 2 tanh: 2 / (1 + exp(-2 x)) - 1; 3 gelu: x / 2 (1 + erf(x / sqrt 2));
// fp32 arithmetic, one rounding to the storage dtype.
// --------------------------------------------------------------------------- //
template <int DT, int VEC>
__global__ __launch_bounds__(256) void activation_kernel(uint16_t* __restrict__ y, const uint16_t* __restrict__ x, int64_t n,
                                                         int kind) {
  const int64_t nvec = n / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    uint16_t xv[VEC], yv[VEC];
    VecIO<VEC>::load(x + i * VEC, xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v = to_f32<DT>(xv[j]);
      float r;
      if (kind == 0) r = v > 0.f ? v : 0.f;
      else if (kind == 1) r = v >= 0.f ? v : to_f32<DT>(mul_storage<DT>(from_f32<DT>(0.01f), xv[j]));
      else if (kind == 2) r = 2.f / (1.f + expf(-2.f * v)) - 1.f;
      else r = v * 0.5f * (1.f + erff(v * 0.70710678118654752440f));
      yv[j] = (kind == 0 && v != v) ? xv[j] : from_f32<DT>(r);  // relu keeps a NaN (tl.maximum propagates it)
    }
    VecIO<VEC>::store(y + i * VEC, yv);
  }
}

extern "C" int ll_activation(void* y, const void* x, int64_t n, int kind, int dtype, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (n < 0 || kind < 0 || kind > 3) return LL_ERR_SHAPE;
  if (!y || !x) return n == 0 ? LL_OK : LL_ERR_ARG;
  if (n == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (n % 8 == 0) && ll_aligned16(y) && ll_aligned16(x);
  const int64_t total = vec ? n / 8 : n;
  const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
#define LL_ACT(DT, VEC) activation_kernel<DT, VEC><<<dim3(grid), 256, 0, st>>>((uint16_t*)y, (const uint16_t*)x, n, kind)
  if (dtype == LL_F16) {
    if (vec) LL_ACT(LL_F16, 8); else LL_ACT(LL_F16, 1);
  } else {
    if (vec) LL_ACT(LL_BF16, 8); else LL_ACT(LL_BF16, 1);
  }
#undef LL_ACT
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// moe_sum -- reference lite_llama/kernels/fused_moe.py:318-335
// --------------------------------------------------------------------------- //
template <int DT, int VEC>
__global__ __launch_bounds__(256) void moe_sum_kernel(uint16_t* __restrict__ out,
                                                      const uint16_t* __restrict__ x,
                                                      int64_t tokens, int top_k, int64_t n) {
  const int64_t nvec = n / VEC;
  const int64_t total = tokens * nvec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / nvec, col = (i % nvec) * VEC;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int kk = 0; kk < top_k; ++kk) {
      uint16_t v[VEC];
      VecIO<VEC>::load(x + (t * top_k + kk) * n + col, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += to_f32<DT>(v[j]);
    }
    uint16_t o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = from_f32<DT>(acc[j]);
    VecIO<VEC>::store(out + t * n + col, o);
  }
}

extern "C" int ll_moe_sum(void* out, const void* x, int64_t tokens, int top_k, int64_t n, int dtype,
                          void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (tokens < 0 || n <= 0 || top_k <= 0) return LL_ERR_SHAPE;
  if (tokens == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (n % 8 == 0) && ll_aligned16(out) && ll_aligned16(x);
  const int64_t total = tokens * (vec ? n / 8 : n);
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  auto oo = (uint16_t*)out;
  auto xx = (const uint16_t*)x;
#define LL_MS(DT, VEC) moe_sum_kernel<DT, VEC><<<dim3(grid), 256, 0, st>>>(oo, xx, tokens, top_k, n)
  if (dtype == LL_F16) {
    if (vec) LL_MS(LL_F16, 8); else LL_MS(LL_F16, 1);
  } else {
    if (vec) LL_MS(LL_BF16, 8); else LL_MS(LL_BF16, 1);
  }
#undef LL_MS
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// rope_emb_forward -- reference lite_llama/kernels/rope_emb.py:14-134
// One block per token; every thread rotates 8-wide (or scalar) channel pairs.
// cos/sin may be fp16/bf16/fp32; only [.., :hd/2] is read.
// --------------------------------------------------------------------------- //
template <int CS>
__device__ __forceinline__ float load_cs(const void* p, int64_t i) {
  if constexpr (CS == LL_F32) return ((const float*)p)[i];
  else return to_f32<CS>(((const uint16_t*)p)[i]);
}

template <int DT, int CS, int VEC>
__global__ __launch_bounds__(256) void rope_kernel(uint16_t* __restrict__ q, uint16_t* __restrict__ k,
                                                   const void* __restrict__ cos_t,
                                                   const void* __restrict__ sin_t, int n_qh, int n_kh,
                                                   int hd, int64_t q_rs, int64_t k_rs, int64_t seq_len,
                                                   int64_t cbs, int64_t css, int64_t sbs, int64_t sss) {
  const int64_t tok = blockIdx.x;
  const int64_t bi = tok / seq_len, si = tok % seq_len;
  const int half = hd / 2;
  const int vph = half / VEC;  // vectors per head-half
  const int total = (n_qh + n_kh) * vph;
  const int64_t cbase = bi * cbs + si * css;
  const int64_t sbase = bi * sbs + si * sss;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / vph, j = (i % vph) * VEC;
    uint16_t* base = (h < n_qh) ? (q + tok * q_rs + (int64_t)h * hd) : (k + tok * k_rs + (int64_t)(h - n_qh) * hd);
    uint16_t x1[VEC], x2[VEC], o1[VEC], o2[VEC];
    VecIO<VEC>::load(base + j, x1);
    VecIO<VEC>::load(base + half + j, x2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float c = load_cs<CS>(cos_t, cbase + j + e);
      const float s = load_cs<CS>(sin_t, sbase + j + e);
      const float a = to_f32<DT>(x1[e]), b = to_f32<DT>(x2[e]);
      o1[e] = from_f32<DT>(a * c - b * s);
      o2[e] = from_f32<DT>(b * c + a * s);
    }
    VecIO<VEC>::store(base + j, o1);
    VecIO<VEC>::store(base + half + j, o2);
  }
}

extern "C" int ll_rope(void* q, void* k, const void* cos_t, const void* sin_t, int64_t tokens,
                       int n_qh, int n_kh, int hd, int64_t q_row_stride, int64_t k_row_stride,
                       int64_t seq_len, int64_t cos_b_stride, int64_t cos_s_stride,
                       int64_t sin_b_stride, int64_t sin_s_stride, int qk_dtype, int cs_dtype,
                       void* stream) {
  if (qk_dtype != LL_F16 && qk_dtype != LL_BF16) return LL_ERR_DTYPE;
  if (cs_dtype != LL_F16 && cs_dtype != LL_BF16 && cs_dtype != LL_F32) return LL_ERR_DTYPE;
  if (tokens < 0 || hd <= 0 || (hd & 1) || seq_len <= 0) return LL_ERR_SHAPE;
  if (tokens == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (hd % 16 == 0) && ll_aligned16(q) && ll_aligned16(k) && (q_row_stride % 8 == 0) &&
                   (k_row_stride % 8 == 0);
  auto qq = (uint16_t*)q;
  auto kk = (uint16_t*)k;
#define LL_ROPE(DT, CS, VEC)                                                                      \
  rope_kernel<DT, CS, VEC><<<dim3((unsigned)tokens), 256, 0, st>>>(                               \
      qq, kk, cos_t, sin_t, n_qh, n_kh, hd, q_row_stride, k_row_stride, seq_len, cos_b_stride,    \
      cos_s_stride, sin_b_stride, sin_s_stride)
#define LL_ROPE_CS(DT, VEC)                                   \
  if (cs_dtype == LL_F16) LL_ROPE(DT, LL_F16, VEC);           \
  else if (cs_dtype == LL_BF16) LL_ROPE(DT, LL_BF16, VEC);    \
  else LL_ROPE(DT, LL_F32, VEC)
  if (qk_dtype == LL_F16) {
    if (vec) { LL_ROPE_CS(LL_F16, 8); } else { LL_ROPE_CS(LL_F16, 1); }
  } else {
    if (vec) { LL_ROPE_CS(LL_BF16, 8); } else { LL_ROPE_CS(LL_BF16, 1); }
  }
#undef LL_ROPE_CS
#undef LL_ROPE
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// rope + KV scatter in one launch (decode step: both are ~1 us of work behind ~2 us of launch
// latency each).  Exactly rope_emb_forward(q, k) followed by update_kv_buffer(cat(k, v)):
// q and k are rotated IN PLACE, and the rotated k heads + the v heads of token i land in pool row
// select_index[i].  kv = [tokens, 2*n_kh, hd] rows (K heads first), token stride given.
// --------------------------------------------------------------------------- //
// CSV: the cos/sin rows are 16-bit and 16-byte aligned -> one vector load each per thread (else
// element loads).  Launched with one thread per (head, vector) so the loop body runs once; every
// load is issued before the first use and none sits in a branch (see skip_rmsnorm_cached).
template <int DT, int CS, int VEC, bool CSV>
__global__ __launch_bounds__(1024) void rope_cache_kernel(
    uint16_t* __restrict__ q, uint16_t* __restrict__ kv, const void* __restrict__ cos_t,
    const void* __restrict__ sin_t, uint16_t* __restrict__ pool, const void* __restrict__ sel, int n_qh,
    int n_kh, int hd, int64_t q_rs, int64_t kv_rs, int64_t seq_len, int64_t cbs, int64_t css, int64_t sbs,
    int64_t sss, int64_t pst, int64_t psh, int idx_w, const int64_t* __restrict__ positions) {
  const int64_t tok = blockIdx.x;
  // positions != NULL: cos/sin are [max_pos, hd] tables indexed by the token's position
  const int64_t bi = positions ? 0 : tok / seq_len, si = positions ? positions[tok] : tok % seq_len;
  const int64_t dst = idx_w == LL_I32 ? (int64_t)((const int32_t*)sel)[tok] : ((const int64_t*)sel)[tok];
  const int half = hd / 2;
  const int vph = half / VEC;  // vectors per head-half
  const int total = (n_qh + 2 * n_kh) * vph;
  const int64_t cbase = bi * cbs + si * css;
  const int64_t sbase = bi * sbs + si * sss;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int h = i / vph, j = (i % vph) * VEC;
    const int hk = h - n_qh;  // >= 0: row of the [2*n_kh, hd] kv block
    uint16_t* base = (hk < 0) ? (q + tok * q_rs + (int64_t)h * hd) : (kv + tok * kv_rs + (int64_t)hk * hd);
    uint16_t x1[VEC], x2[VEC], o1[VEC], o2[VEC];
    VecIO<VEC>::load(base + j, x1);
    VecIO<VEC>::load(base + half + j, x2);
    float c[VEC], sn[VEC];
    if constexpr (CSV) {
      uint16_t cv[VEC], sv[VEC];
      VecIO<VEC>::load((const uint16_t*)cos_t + cbase + j, cv);
      VecIO<VEC>::load((const uint16_t*)sin_t + sbase + j, sv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        c[e] = to_f32<CS == LL_F32 ? LL_F16 : CS>(cv[e]);
        sn[e] = to_f32<CS == LL_F32 ? LL_F16 : CS>(sv[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        c[e] = load_cs<CS>(cos_t, cbase + j + e);
        sn[e] = load_cs<CS>(sin_t, sbase + j + e);
      }
    }
    const bool rot = hk < n_kh;  // q and k heads rotate, v heads pass through
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float a = to_f32<DT>(x1[e]), b = to_f32<DT>(x2[e]);
      const uint16_t r1 = from_f32<DT>(a * c[e] - b * sn[e]);
      const uint16_t r2 = from_f32<DT>(b * c[e] + a * sn[e]);
      o1[e] = rot ? r1 : x1[e];
      o2[e] = rot ? r2 : x2[e];
    }
    if (rot) {
      VecIO<VEC>::store(base + j, o1);
      VecIO<VEC>::store(base + half + j, o2);
    }
    if (hk >= 0) {
      uint16_t* prow = pool + dst * pst + (int64_t)hk * psh;
      VecIO<VEC>::store(prow + j, o1);
      VecIO<VEC>::store(prow + half + j, o2);
    }
  }
}

extern "C" int ll_rope_kv_update(void* q, void* kv, const void* cos_t, const void* sin_t, void* kv_buffer,
                                 const void* select_index, int64_t tokens, int n_qh, int n_kh, int hd,
                                 int64_t q_row_stride, int64_t kv_row_stride, int64_t seq_len,
                                 int64_t cos_b_stride, int64_t cos_s_stride, int64_t sin_b_stride,
                                 int64_t sin_s_stride, int64_t pool_stride_t, int64_t pool_stride_h,
                                 int qk_dtype, int cs_dtype, int idx_width, const int64_t* positions,
                                 void* stream) {
  if (qk_dtype != LL_F16 && qk_dtype != LL_BF16) return LL_ERR_DTYPE;
  if (cs_dtype != LL_F16 && cs_dtype != LL_BF16 && cs_dtype != LL_F32) return LL_ERR_DTYPE;
  if (idx_width != LL_I32 && idx_width != LL_I64) return LL_ERR_DTYPE;
  if (tokens < 0 || hd <= 0 || (hd & 1) || seq_len <= 0 || n_qh < 0 || n_kh <= 0) return LL_ERR_SHAPE;
  if (tokens == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (hd % 16 == 0) && ll_aligned16(q) && ll_aligned16(kv) && ll_aligned16(kv_buffer) &&
                   (q_row_stride % 8 == 0) && (kv_row_stride % 8 == 0) && (pool_stride_t % 8 == 0) &&
                   (pool_stride_h % 8 == 0);
  const bool csv = vec && cs_dtype != LL_F32 && ll_aligned16(cos_t) && ll_aligned16(sin_t) &&
                   ((cos_b_stride | cos_s_stride | sin_b_stride | sin_s_stride) % 8 == 0);
  const int work = (n_qh + 2 * n_kh) * (hd / 2 / (vec ? 8 : 1));
  const unsigned threads = (unsigned)(work >= 1024 ? 1024 : (work + 63) / 64 * 64);
#define LL_RC(DT, CS, VEC, CSV)                                                                         \
  rope_cache_kernel<DT, CS, VEC, CSV><<<dim3((unsigned)tokens), threads, 0, st>>>(                      \
      (uint16_t*)q, (uint16_t*)kv, cos_t, sin_t, (uint16_t*)kv_buffer, select_index, n_qh, n_kh, hd,    \
      q_row_stride, kv_row_stride, seq_len, cos_b_stride, cos_s_stride, sin_b_stride, sin_s_stride,     \
      pool_stride_t, pool_stride_h, idx_width, positions)
#define LL_RC_CS(DT, VEC)                                                      \
  if (cs_dtype == LL_F16) {                                                    \
    if (csv) LL_RC(DT, LL_F16, VEC, true); else LL_RC(DT, LL_F16, VEC, false);   \
  } else if (cs_dtype == LL_BF16) {                                            \
    if (csv) LL_RC(DT, LL_BF16, VEC, true); else LL_RC(DT, LL_BF16, VEC, false); \
  } else LL_RC(DT, LL_F32, VEC, false)
  if (qk_dtype == LL_F16) {
    if (vec) { LL_RC_CS(LL_F16, 8); } else { LL_RC_CS(LL_F16, 1); }
  } else {
    if (vec) { LL_RC_CS(LL_BF16, 8); } else { LL_RC_CS(LL_BF16, 1); }
  }
#undef LL_RC_CS
#undef LL_RC
  return LL_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// greedy argmax -- reference lite_llama/engine/sampler.py:227-228,264
// (torch.argmax: first index of the maximum).  One block per row.
// --------------------------------------------------------------------------- //
template <int DT>
__device__ __forceinline__ float load_logit(const void* p, int64_t i) {
  if constexpr (DT == LL_F32) return ((const float*)p)[i];
  else return to_f32<DT>(((const uint16_t*)p)[i]);
}

template <int DT>
__global__ __launch_bounds__(1024) void argmax_kernel(int64_t* __restrict__ out,
                                                      const void* __restrict__ logits, int64_t n,
                                                      int64_t stride_row) {
  __shared__ float sv[16];
  __shared__ int64_t si[16];
  const int64_t row = blockIdx.x;
  float best = -INFINITY;
  int64_t bi = INT64_MAX;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = load_logit<DT>(logits, row * stride_row + i);
    if (bi == INT64_MAX || v > best) {  // per-thread indices ascend: strict > keeps the first max
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int64_t oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[wid] = best;
    si[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    out[row] = bi;
  }
}

// Two-stage variant for vocabulary-sized rows: a [rows x chunks] grid fills the chip (one block
// per row leaves 3/4 of the CUs idle at batch 64), a second tiny kernel merges the per-chunk
// winners in chunk order (strict >, so the first maximum still wins).
struct ArgPart {
  float v;
  int pad;
  int64_t i;
};

template <int DT>
__global__ __launch_bounds__(256) void argmax_part_kernel(ArgPart* __restrict__ part,
                                                          const void* __restrict__ logits, int64_t n,
                                                          int64_t stride_row, int chunks) {
  __shared__ float sv[4];
  __shared__ int64_t si[4];
  const int64_t row = blockIdx.y;
  const int64_t per = (n + chunks - 1) / chunks;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  float best = -INFINITY;
  int64_t bi = INT64_MAX;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = load_logit<DT>(logits, row * stride_row + i);
    if (bi == INT64_MAX || v > best) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int64_t oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[wid] = best;
    si[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    ArgPart o;
    o.v = best;
    o.pad = 0;
    o.i = bi;
    part[row * chunks + blockIdx.x] = o;
  }
}

__global__ __launch_bounds__(64) void argmax_merge_kernel(int64_t* __restrict__ out,
                                                         const ArgPart* __restrict__ part, int chunks) {
  const int64_t row = blockIdx.x;
  float best = -INFINITY;
  int64_t bi = INT64_MAX;
  for (int c = threadIdx.x; c < chunks; c += 64) {
    const ArgPart pp = part[row * chunks + c];
    if (pp.i != INT64_MAX && (bi == INT64_MAX || pp.v > best)) {
      best = pp.v;
      bi = pp.i;
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
  if (threadIdx.x == 0) out[row] = bi;
}

extern "C" int ll_argmax_split(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t stride_row,
                               int dtype, void* scratch, int chunks, void* stream) {
  if (rows < 0 || n <= 0 || chunks <= 0 || chunks > 65535) return LL_ERR_SHAPE;
  if (!scratch) return LL_ERR_ARG;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)chunks, (unsigned)rows);
  ArgPart* part = (ArgPart*)scratch;
  if (dtype == LL_F16) argmax_part_kernel<LL_F16><<<grid, 256, 0, st>>>(part, logits, n, stride_row, chunks);
  else if (dtype == LL_BF16) argmax_part_kernel<LL_BF16><<<grid, 256, 0, st>>>(part, logits, n, stride_row, chunks);
  else if (dtype == LL_F32) argmax_part_kernel<LL_F32><<<grid, 256, 0, st>>>(part, logits, n, stride_row, chunks);
  else return LL_ERR_DTYPE;
  argmax_merge_kernel<<<dim3((unsigned)rows), 64, 0, st>>>(out, part, chunks);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_argmax(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t stride_row,
                         int dtype, void* stream) {
  if (rows < 0 || n <= 0) return LL_ERR_SHAPE;
  if (rows == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16) argmax_kernel<LL_F16><<<dim3((unsigned)rows), 1024, 0, st>>>(out, logits, n, stride_row);
  else if (dtype == LL_BF16) argmax_kernel<LL_BF16><<<dim3((unsigned)rows), 1024, 0, st>>>(out, logits, n, stride_row);
  else if (dtype == LL_F32) argmax_kernel<LL_F32><<<dim3((unsigned)rows), 1024, 0, st>>>(out, logits, n, stride_row);
  else return LL_ERR_DTYPE;
  return LL_LAUNCH_CHECK();
}
