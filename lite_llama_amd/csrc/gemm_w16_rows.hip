// Decode-shaped F.linear for UNQUANTISED fp16 / bf16 weights [N][K] (M <= 64 rows): the row-group loop of gemm_w4_v4.hip without
// a dequantiser.  Reference: lite_llama/models/quantization/methods/unquantized.py:21-22 and the lm_head of models/base.py:486-489
// are torch's library GEMM; BASELINE config 2 (Qwen2.5-1.5B bf16, batch 32) has its fused gate|up and the lm_head there.
//   * A workgroup owns a contiguous run of 32-row groups (N / 32 groups dealt over the CUs, sizes differing by at most one) and ALL
//     of K; it walks its run in tiles of up to 4 groups.  Finished outputs only: no split-K planes, no finish launch, bias or the
//     fused swiglu of a row-interleaved gate|up in the epilogue (the decode engines' epilogue, common.h::ll_silu_mul_f32).
//   * 8 waves: four LOADERS stream, per 128-k chunk, the tile's 4 x [32 rows x 256 B] weight blocks and the [M x 256 B] activation
//     block into an R-slot LDS ring by LDS-DMA -- row-major with the 16-byte slots of row r at slot ^ (r & 15), swizzled through the
//     per-lane SOURCE address (the activation image of the int4 engines, now for both operands): fragment reads are conflict free;
//     weights non-temporal.  Four CONSUMERS, one per row group of the tile, run the chunk's 8 x MT MFMA 32x32x16 from LDS
//     fragments and keep the tile's accumulators (16 x MT registers) for the whole K loop: no cross-wave exchange.  The loaders
//     run ahead ACROSS tile boundaries, so a tile's epilogue overlaps the next tile's first chunks.
//   * one s_barrier per chunk; the loaders' operation count per chunk is static (8 + 2 MT pieces each).
#include <stdlib.h>

#include "common.h"
#include "gemm_w4_common.h"

typedef __bf16 r16_bf16x8 __attribute__((ext_vector_type(8)));

#define R16_THREADS 512
template <int MT>
struct R16Lds {
  static constexpr int W_BYTES = 4 * 8192;        // four row groups x [32 rows][256 B]
  static constexpr int X_BYTES = MT * 8192;       // [32 MT rows][256 B]
  static constexpr int SLOT = W_BYTES + X_BYTES;  // 40 KB (M <= 32) / 48 KB
  static constexpr int R = MT == 1 ? 4 : 3;
  static constexpr int BYTES = R * SLOT;
  static_assert(BYTES <= 160 * 1024, "LDS");
};

struct R16Params {
  uint16_t* out;
  const uint16_t* x;
  const uint16_t* w;
  const uint16_t* bias;
  int64_t m, n, k, x_stride, w_stride;
  int chunks;        // K / 128
  int rbase, rrem;   // workgroup b owns row groups [b * rbase + min(b, rrem), + rbase + (b < rrem))
  int epi;           // 0: out[m, n] (+ bias);  1: rows are (gate_j, up_j) pairs -> out[m, n / 2] = silu(gate) * up
  const float* a_scale;  // R16_I8 only: per-token activation scales [m], per-channel weight scales [n] (smoothquant, w8a8.py:118-149)
  const float* w_scale;
};
#define R16_I8 2  // DT code of the int8 x int8 form: x / w are int8 rows of 256 k per chunk, out is fp16

template <int N>
__device__ __forceinline__ void r16_wait_units(int k) {  // at most k units of N operations each may still be in flight
  if (k <= 0) v3_vmcnt<0>();
  else if (k == 1) v3_vmcnt<N>();
  else v3_vmcnt<2 * N>();
}

template <int MT, int DT>
__global__ __launch_bounds__(R16_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgemm16_rows_kernel(const R16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using LD = R16Lds<MT>;
  constexpr int R = LD::R;
  constexpr int XP = 2 * MT;   // activation pieces (4 rows x 256 B) per loader wave
  constexpr int OPS = 8 + XP;  // LDS-DMA operations per loader wave and chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (int)blockIdx.x;
  const int lo = b * p.rbase + (b < p.rrem ? b : p.rrem);
  const int cnt = p.rbase + (b < p.rrem ? 1 : 0);
  if (cnt <= 0) return;
  const int hi = lo + cnt;
  const int tiles = (cnt + 3) >> 2;
  const int units = tiles * p.chunks;

  if (wv >= 4) {
    // ======================================== loaders ======================================== //
    const int L = wv - 4;  // weight block of row group L of every tile + activation pieces XP * L ..
    uint32_t woff[8], xoff[XP];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = j * 4 + (lane >> 4), slot = (lane & 15) ^ (r & 15);
      woff[j] = (uint32_t)((int64_t)r * p.w_stride * 2 + slot * 16);
    }
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      int64_t r = (XP * L + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (int)(r & 15);
      if (r >= p.m) r = p.m - 1;  // rows >= M feed only unstored outputs
      xoff[j] = (uint32_t)(r * p.x_stride * 2 + slot * 16);
    }
    int issued = 0, t = 0, c = 0;
    uint32_t dst = 0;
    auto issue = [&]() {
      const int rg = lo + 4 * t + L;
      // a short last tile: the idle loader keeps its eight operations (the counted waits stay static) but points them at ONE
      // cached KB (the first of the matrix) -- no weight bytes are fetched twice; its LDS block is never read
      const bool live = rg < hi;
      const char* wb = live ? (const char*)p.w + (size_t)rg * 32 * (size_t)p.w_stride * 2 + (size_t)c * 256 : (const char*)p.w;
      const char* xb = (const char*)p.x + (size_t)c * 256;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (live) v3_dma16<true>(dst + L * 8192 + j * 1024, wb, woff[j]);
        else v3_dma16<false>(dst + L * 8192 + j * 1024, wb, (uint32_t)(lane * 16));
      }
#pragma unroll
      for (int j = 0; j < XP; ++j) v3_dma16<false>(dst + LD::W_BYTES + (XP * L + j) * 1024, xb, xoff[j]);
      ++issued;
      if (++c == p.chunks) { c = 0; ++t; }
      dst = dst + LD::SLOT == R * LD::SLOT ? 0 : dst + LD::SLOT;
    };
    const int pre = units < R - 1 ? units : R - 1;
    for (int i = 0; i < pre; ++i) issue();
    r16_wait_units<OPS>(issued - 1);  // unit 0 has landed
    asm volatile("s_barrier" ::: "memory");  // P0
    for (int u = 0; u < units; ++u) {
      if (issued < units) issue();  // into the slot of unit u - 1, free since the barrier that ended it
      const int need = units < u + 2 ? units : u + 2;
      r16_wait_units<OPS>(issued - need);  // unit u + 1 has landed before the consumers are released into it
      asm volatile("s_barrier" ::: "memory");  // B_u
    }
    return;
  }

  // ======================================= consumers ======================================= //
  const int nl = lane & 31, h = lane >> 5;
  const int wrow = wv * 8192 + nl * 256;  // this wave's row group of the tile, weight row nl
  constexpr int ODT = DT == R16_I8 ? LL_F16 : DT;  // output / bias storage type
  f32x16 acc[MT];
  i32x16 iacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[mt][e] = 0.f; iacc[mt][e] = 0; }
  auto swap32 = [](uint32_t& a, uint32_t& bb) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, bb, false, false);
    a = r[0];
    bb = r[1];
  };
  const bool has_bias = p.bias != nullptr;
  int t = 0, c = 0;
  uint32_t slot = 0;
  asm volatile("s_barrier" ::: "memory");  // P0: unit 0 has landed
  for (int u = 0; u < units; ++u) {
    const int rg = lo + 4 * t + wv;
    const bool mine = rg < hi;
    if (mine) {
      const unsigned char* wb = lds + slot + wrow;
      const unsigned char* xb = lds + slot + LD::W_BYTES + nl * 256;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int off = ((2 * s + h) ^ (nl & 15)) * 16;
        const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + off);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const u32x4 xf = *reinterpret_cast<const u32x4*>(xb + mt * 8192 + off);
          if constexpr (DT == LL_F16)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), acc[mt], 0, 0, 0);
          else if constexpr (DT == LL_BF16)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(r16_bf16x8, wf), __builtin_bit_cast(r16_bf16x8, xf), acc[mt], 0, 0, 0);
          else  // 16 int8 per lane and k-step of 32: the same 16-byte slots, exact int32 sums
            iacc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wf), __builtin_bit_cast(i32x4, xf), iacc[mt], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B_u: this unit's slot may be refilled
    slot = slot + LD::SLOT == R * LD::SLOT ? 0 : slot + LD::SLOT;
    if (++c == p.chunks) {
      c = 0;
      if (mine) {
        // a lane holds, for batch row nl (+ 32 mt), the weight rows 8g + 4h .. + 3 of the group: the decode engines' epilogue
        const int64_t ncol = (int64_t)rg * 32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int64_t mrow = nl + mt * 32;
          const bool row_ok = mrow < p.m;
          float v[16];
          if constexpr (DT == R16_I8) {
            // (acc.f32 * a_scale[m]) * w_scale[n], each product rounded to fp32 before the next (w8a8.py:118-120): pinned so that
            // the narrowing conversion below cannot be fused into it (common.h::ll_silu_mul_f32 has the story)
            const float as = p.a_scale[row_ok ? mrow : 0];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 ws = *reinterpret_cast<const f32x4*>(p.w_scale + ncol + 8 * g + 4 * h);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float f = ((float)iacc[mt][4 * g + e] * as) * ws[e];
                asm volatile("" : "+v"(f));
                v[4 * g + e] = f;
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[mt][e];
          }
          if (has_bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ncol + 8 * g + 4 * h);
              v[4 * g + 0] += to_f32<ODT>((uint16_t)(bb.x & 0xffffu));
              v[4 * g + 1] += to_f32<ODT>((uint16_t)(bb.x >> 16));
              v[4 * g + 2] += to_f32<ODT>((uint16_t)(bb.y & 0xffffu));
              v[4 * g + 3] += to_f32<ODT>((uint16_t)(bb.y >> 16));
            }
          }
          uint32_t lo2[4], hi2[4], sw[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float f = v[4 * g + e];
              asm volatile("" : "+v"(f));
              o[e] = from_f32<ODT>(f);
            }
            lo2[g] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            hi2[g] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            if (p.epi)  // both outputs rounded to the storage dtype, then silu(g) * u in fp32: swiglu_forward's arithmetic
              sw[g] = (uint32_t)from_f32<ODT>(ll_silu_mul_f32(to_f32<ODT>(o[0]), to_f32<ODT>(o[1]))) |
                      ((uint32_t)from_f32<ODT>(ll_silu_mul_f32(to_f32<ODT>(o[2]), to_f32<ODT>(o[3]))) << 16);
          }
          const int64_t nn = ncol + 16 * h;
          if (p.epi) {
            swap32(sw[0], sw[2]);
            swap32(sw[1], sw[3]);
            if (row_ok) *reinterpret_cast<u32x4*>(p.out + mrow * (p.n >> 1) + (nn >> 1)) = u32x4{sw[0], sw[2], sw[1], sw[3]};
          } else {
            swap32(lo2[0], lo2[2]);
            swap32(hi2[0], hi2[2]);
            swap32(lo2[1], lo2[3]);
            swap32(hi2[1], hi2[3]);
            if (row_ok) {
              *reinterpret_cast<u32x4*>(p.out + mrow * p.n + nn) = u32x4{lo2[0], hi2[0], lo2[2], hi2[2]};
              *reinterpret_cast<u32x4*>(p.out + mrow * p.n + nn + 8) = u32x4{lo2[1], hi2[1], lo2[3], hi2[3]};
            }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) { acc[mt][e] = 0.f; iacc[mt][e] = 0; }
        }
      }
      ++t;
    }
  }
}

static int r16_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// the row-group loop needs R16Lds<MT>::BYTES of LDS per workgroup (160 KB at MT = 1): callers fall back when the device's opt-in
// limit is smaller
static bool r16_lds_fits(int64_t m) {
  static int optin[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;  // no device (host-side plan queries): gfx950 assumed
  if (!optin[dev]) {
    int v = 0;
    optin[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && v > 0 ? v : 160 * 1024;
  }
  return (m <= 32 ? R16Lds<1>::BYTES : R16Lds<2>::BYTES) <= optin[dev];
}

extern "C" int ll_dense16_rows_supported(int64_t m, int64_t n, int64_t k, int epilogue) {
  if (m < 1 || m > 64 || n < 32 || n % 32 != 0 || k < 128 || k % 128 != 0) return 0;
  if (epilogue != 0 && epilogue != 1) return 0;
  return r16_lds_fits(m) ? 1 : 0;
}

// out [m][n] (epilogue 0, + bias) or [m][n / 2] (epilogue 1: rows of w interleaved (gate_j, up_j)) = x [m][k] @ w [n][k]^T for an
// unquantised fp16 / bf16 weight; m <= 64, n % 32 == 0, k % 128 == 0, rows 16-byte aligned.  Returns LL_OK, or an error code.
extern "C" int ll_dense16_rows_matmul(void* out, const void* x, const void* w, const void* bias, int64_t m, int64_t n, int64_t k,
                                      int64_t x_stride_m, int64_t w_stride_n, int dtype, int epilogue, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (m < 0 || n <= 0 || k <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  if (!ll_dense16_rows_supported(m, n, k, epilogue) || x_stride_m % 8 != 0 || w_stride_n % 8 != 0) return LL_ERR_SHAPE;
  if (!out || !x || !w || !ll_aligned16(out) || !ll_aligned16(x) || !ll_aligned16(w) || (bias && ((uintptr_t)bias & 7))) return LL_ERR_ARG;
  if ((m - 1) * x_stride_m * 2 + k * 2 >= (1ll << 31) || 32 * w_stride_n * 2 >= (1ll << 31)) return LL_ERR_SHAPE;
  R16Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m; p.w_stride = w_stride_n;
  p.chunks = (int)(k / 128);
  p.epi = epilogue;
  const int64_t rgs = n / 32;
  int grid = r16_num_cus();
  if (grid > rgs) grid = (int)rgs;
  p.rbase = (int)(rgs / grid); p.rrem = (int)(rgs % grid);
  hipStream_t st = (hipStream_t)stream;
#define R16_GO(MT, DT)                                                                                                   \
  {                                                                                                                      \
    static bool attr_[16] = {false}; /* per device */                                                   \
    int dev_ = 0;                                                                                                        \
    (void)hipGetDevice(&dev_);                                                                                           \
    if (dev_ >= 0 && dev_ < 16 && !attr_[dev_]) {                                                                        \
      (void)hipFuncSetAttribute((const void*)wgemm16_rows_kernel<MT, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, R16Lds<MT>::BYTES); \
      attr_[dev_] = true;                                                                                                \
    }                                                                                                                    \
    wgemm16_rows_kernel<MT, DT><<<dim3((unsigned)grid), R16_THREADS, R16Lds<MT>::BYTES, st>>>(p);                          \
  }
  if (m <= 32) {
    if (dtype == LL_F16) R16_GO(1, LL_F16) else R16_GO(1, LL_BF16)
  } else {
    if (dtype == LL_F16) R16_GO(2, LL_F16) else R16_GO(2, LL_BF16)
  }
#undef R16_GO
  return LL_LAUNCH_CHECK();
}

// smoothquant (W8A8) form: out [m][n] fp16 (epilogue 0, + bias) or [m][n / 2] (epilogue 1: rows of qw interleaved (gate_j, up_j)) =
// fp16(((float)(qa [m][k] int8 @ qw [n][k]^T int32) * a_scale[m]) * w_scale[n] (+ bias)) -- ll_w8a8_matmul's epilogue
// (kernels/quantization/w8a8.py:118-149), exact int32 sums on mfma_i32_32x32x32_i8; then silu(gate) * up on the rounded outputs
// (= ll_w8a8_finish_swiglu).  m <= 64, n % 32 == 0, k % 256 == 0, row strides (bytes) % 16 == 0.
extern "C" int ll_w8a8_rows_supported(int64_t m, int64_t n, int64_t k, int epilogue) {
  return (m >= 1 && m <= 64 && n >= 32 && n % 32 == 0 && k >= 256 && k % 256 == 0 && (epilogue == 0 || epilogue == 1) && r16_lds_fits(m)) ? 1 : 0;
}

extern "C" int ll_w8a8_rows_matmul(void* out, const int8_t* qa, const float* a_scale, const int8_t* qw, const float* w_scale,
                                   const void* bias, int64_t m, int64_t n, int64_t k, int64_t qa_stride_m, int64_t qw_stride_n,
                                   int epilogue, void* stream) {
  if (m < 0 || n <= 0 || k <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  if (!ll_w8a8_rows_supported(m, n, k, epilogue) || qa_stride_m % 16 != 0 || qw_stride_n % 16 != 0) return LL_ERR_SHAPE;
  if (!out || !qa || !qw || !a_scale || !w_scale || !ll_aligned16(out) || !ll_aligned16(qa) || !ll_aligned16(qw) || !ll_aligned16(w_scale) ||
      (bias && ((uintptr_t)bias & 7)))
    return LL_ERR_ARG;
  if ((m - 1) * qa_stride_m + k >= (1ll << 31) || 32 * qw_stride_n >= (1ll << 31)) return LL_ERR_SHAPE;
  R16Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)qa; p.w = (const uint16_t*)qw; p.bias = (const uint16_t*)bias;
  p.a_scale = a_scale; p.w_scale = w_scale;
  p.m = m; p.n = n; p.k = k;
  p.x_stride = qa_stride_m / 2; p.w_stride = qw_stride_n / 2;  // the kernel addresses rows in 2-byte units
  p.chunks = (int)(k / 256);                                    // 256 bytes of a row per chunk
  p.epi = epilogue;
  const int64_t rgs = n / 32;
  int grid = r16_num_cus();
  if (grid > rgs) grid = (int)rgs;
  p.rbase = (int)(rgs / grid); p.rrem = (int)(rgs % grid);
  hipStream_t st = (hipStream_t)stream;
#define R16_GO8(MT)                                                                                                      \
  {                                                                                                                      \
    static bool attr_[16] = {false}; /* per device */                                                   \
    int dev_ = 0;                                                                                                        \
    (void)hipGetDevice(&dev_);                                                                                           \
    if (dev_ >= 0 && dev_ < 16 && !attr_[dev_]) {                                                                        \
      (void)hipFuncSetAttribute((const void*)wgemm16_rows_kernel<MT, R16_I8>, hipFuncAttributeMaxDynamicSharedMemorySize, R16Lds<MT>::BYTES); \
      attr_[dev_] = true;                                                                                                \
    }                                                                                                                    \
    wgemm16_rows_kernel<MT, R16_I8><<<dim3((unsigned)grid), R16_THREADS, R16Lds<MT>::BYTES, st>>>(p);                      \
  }
  if (m <= 32) R16_GO8(1) else R16_GO8(2)
#undef R16_GO8
  return LL_LAUNCH_CHECK();
}
