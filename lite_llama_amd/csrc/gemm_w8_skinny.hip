// Decode-shaped (M <= 64) GEMMs over 8-bit weights -- a9 w8a16_matmul (lite_llama/kernels/quantization/w8a16.py:28-207: fp8
// e4m3 / int8 weights, one fp32 scale per group_n x group_k block, fp16 activations) and a10 smoothquant_matmul
// (w8a8.py:28-217: per-token int8 activations x per-channel int8 weights, exact int32 accumulation) -- as a split-K
// weight-streaming kernel in the form of moe.hip::moe_gemm_kernel2: the 128 x 128 weight tile of a chunk is fetched with
// FULL-LINE loads (8 lanes per 128-byte row, the next chunk's loads in flight under the current chunk's MFMAs) and handed
// to the MFMA layout through LDS.  The generic engine (gemm_wq.hip::wgemm_kernel) lets every lane pick 16-byte pieces
// out of its own weight row at a K-byte stride and reaches 1.4 TB/s on the Llama-3-8B W8A8 shapes; it keeps every shape
// this kernel does not take (M > 64, K % 128 != 0, odd scale groups).
//
// grid = (N / 128 tiles, S splits): a workgroup owns chunks [y * cps, (y + 1) * cps) of its tile and leaves an fp32 (int32
// for a10) partial [S][M][N]; dense8_finish adds the S partials in split order and applies the epilogue (a9: + bias;
// a10: * a_scale[m] * w_scale[n] + bias, optionally the raw int32 sums) -- integer sums are exact in any order, the
// fp32 sums are rounded once per split and once at the end.
#include <stdlib.h>

#include "common.h"
#include "gemm_w8_common.h"

namespace {

struct alignas(16) Q4 {
  uint32_t x, y, z, w;
};

constexpr int D8_FP8 = 1, D8_I8 = 2, D8_I8I8 = 3, D8_F16 = 4, D8_BF16 = 5;  // 4 / 5: unquantised 16-bit weights (round 3)

struct D8Params {
  void* part;             // [S][M][N] fp32 (a9) or int32 (a10)
  const void* x;          // fp16 [M, K] (a9) or int8 [M, K] (a10), row stride x_stride elements
  const unsigned char* w; // [N, K] bytes, row stride w_stride
  const float* scales;    // a9: [ceil(N / gn)][ceil(K / gk)] block scales (strides s_stride_n / s_stride_k)
  int64_t m, n, k, x_stride, w_stride, s_stride_n, s_stride_k, group_k;
  int group_n, cps, chunks;
  // one split only (16-bit form): the kernel applies the epilogue itself -- out 16-bit [M][N], bias or nullptr
  uint16_t* direct_out;
  const uint16_t* direct_bias;
  int nt_w;               // weight stream read with the non-temporal policy (d8_nt: big, bandwidth-bound streams only)
};
// Non-temporal weight loads pay where the launch is bandwidth-bound -- lm_head 151936 x 1536 fp16: 97.3 -> 88.7 us, 152064 x
// 3584: 230 -> 220, Llama-3-8B W8A8 step 4.92 -> 4.76 ms with nt everywhere -- and COST 4 - 6 % on mid-size streams that run at
// ~2.4 TB/s (17920 x 1536 fp16: 22.5 -> 24.1 us; same box, round 3): by size, LL_D8_NT_MIN_MB (default 96) megabytes.
static int d8_nt(int64_t weight_bytes) {
  static const int64_t min_mb = getenv("LL_D8_NT_MIN_MB") ? atoll(getenv("LL_D8_NT_MIN_MB")) : 96;
  return weight_bytes >= min_mb * (1ll << 20) ? 1 : 0;
}

typedef __bf16 d8_bf16x8 __attribute__((ext_vector_type(8)));

template <int WFMT, int MT>
__global__ __launch_bounds__(256) void dense8_kernel(const D8Params p) {
  constexpr bool I8A = WFMT == D8_I8I8;
  constexpr bool W16 = WFMT == D8_F16 || WFMT == D8_BF16;  // 16-bit weights: a chunk is 128 rows x 64 k (still 128 B per row)
  constexpr int KCH = W16 ? 64 : 128;                   // k per chunk
  constexpr int AB = I8A ? 1 : 2;                       // bytes per activation element
  constexpr int A_ROW_BYTES = KCH * AB + 16, W_ROW_BYTES = 128 + 16;
  constexpr int AP = KCH * AB / 16;                      // 16-byte pieces per activation row and chunk
  constexpr int APASS = (MT * 32 * AP + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[MT * 32 * A_ROW_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char lds_w[128 * W_ROW_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nl = lane & 31, h = lane >> 5;
  const int64_t ntile = (int64_t)blockIdx.x * 128;
  const int c_lo = blockIdx.y * p.cps, c_hi = c_lo + p.cps < p.chunks ? c_lo + p.cps : p.chunks;
  if (c_lo >= c_hi) return;  // (never: the host sizes the splits so that every one has a chunk)

  const unsigned char* wsrc[4];
  int wdst[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int i = ps * 256 + tid, row = i >> 3, q = i & 7;
    int64_t nrow = ntile + row;
    if (nrow >= p.n) nrow = p.n - 1;                     // rows past N feed outputs that are never stored
    wsrc[ps] = p.w + nrow * p.w_stride + q * 16;
    wdst[ps] = row * W_ROW_BYTES + q * 16;
  }
  const unsigned char* asrc[APASS];
  int adst[APASS];
  bool aok[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int i = ps * 256 + tid, row = i / AP, q = i % AP;
    aok[ps] = i < MT * 32 * AP && row < p.m;
    asrc[ps] = (const unsigned char*)p.x + ((int64_t)(aok[ps] ? row : 0) * p.x_stride) * AB + q * 16;
    adst[ps] = (i < MT * 32 * AP ? row : 0) * A_ROW_BYTES + q * 16;
  }
  int64_t crow = ntile + wv * 32 + nl;
  if (crow >= p.n) crow = p.n - 1;
  const float* srow = (I8A || W16) ? nullptr : p.scales + (crow / p.group_n) * p.s_stride_n;
  // 8-bit: lane (nl, h) owns the k-half h of the chunk; 16-bit: MFMA step s of the chunk reads k = 16 s + 8 h .. + 8
  const unsigned char* wfrag_base = lds_w + (wv * 32 + nl) * W_ROW_BYTES + (W16 ? h * 16 : h * 64);
  const unsigned char* afrag_base = lds_a + nl * A_ROW_BYTES + (W16 ? h * 16 : h * (64 * AB));

  f32x16 accf[MT];
  i32x16 acci[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accf[mt][r] = 0.f;
      acci[mt][r] = 0;
    }

  i32x4 wreg[4], areg[APASS];
  auto fetch = [&](int c) {
    if (p.nt_w) {  // (wave-uniform) the weight stream: whole 128-byte lines, read once
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) wreg[ps] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wsrc[ps] + (int64_t)c * 128));
    } else {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) wreg[ps] = *reinterpret_cast<const i32x4*>(wsrc[ps] + (int64_t)c * 128);
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps)  // rows without a token all re-read ONE piece: their LDS rows are zeroed
      areg[ps] = *reinterpret_cast<const i32x4*>(aok[ps] ? asrc[ps] + (int64_t)c * (KCH * AB) : (const unsigned char*)p.x);
  };
  fetch(c_lo);
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<i32x4*>(lds_w + wdst[ps]) = wreg[ps];
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps)
      if (ps * 256 + tid < MT * 32 * AP) *reinterpret_cast<i32x4*>(lds_a + adst[ps]) = aok[ps] ? areg[ps] : i32x4{0, 0, 0, 0};
    __syncthreads();
    fetch(c + 1 < c_hi ? c + 1 : c);  // in flight under this chunk's arithmetic (unconditional: no branch around loads)
    if constexpr (I8A) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const i32x4 wfrag = *reinterpret_cast<const i32x4*>(wfrag_base + s * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const i32x4 afrag = *reinterpret_cast<const i32x4*>(afrag_base + mt * 32 * A_ROW_BYTES + s * 16);
          acci[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfrag, afrag, acci[mt], 0, 0, 0);
        }
      }
    } else if constexpr (W16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const i32x4 wraw = *reinterpret_cast<const i32x4*>(wfrag_base + s * 32);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const i32x4 araw = *reinterpret_cast<const i32x4*>(afrag_base + mt * 32 * A_ROW_BYTES + s * 32);
          if constexpr (WFMT == D8_F16)
            accf[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wraw), __builtin_bit_cast(f16x8, araw), accf[mt], 0, 0, 0);
          else
            accf[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(d8_bf16x8, wraw), __builtin_bit_cast(d8_bf16x8, araw), accf[mt], 0, 0, 0);
        }
      }
    } else {
      const int kbase = c * 128 + h * 64;
      const float sv = srow[(int64_t)(kbase / p.group_k) * p.s_stride_k];
      const uint32_t sp = d8_bcast(WFMT == D8_FP8 ? sv * 256.0f : sv);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint2 raw = *reinterpret_cast<const uint2*>(wfrag_base + s * 8);
        Q4 wf;
        if constexpr (WFMT == D8_FP8) {
          d8_fp8(raw.x, sp, wf.x, wf.y);
          d8_fp8(raw.y, sp, wf.z, wf.w);
        } else {
          d8_i8(raw.x, sp, wf.x, wf.y);
          d8_i8(raw.y, sp, wf.z, wf.w);
        }
        const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f16x8 afrag = *reinterpret_cast<const f16x8*>(afrag_base + mt * 32 * A_ROW_BYTES + s * 16);
          accf[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, afrag, accf[mt], 0, 0, 0);
        }
      }
    }
  }

  if constexpr (W16) {
    if (p.direct_out) {  // a single split: fp32 sums -> (+ bias) -> one rounding, stored straight to the output
      constexpr int DT = WFMT == D8_F16 ? LL_F16 : LL_BF16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + nl;
        if (m >= p.m) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t nn = ntile + wv * 32 + 8 * g + 4 * h;
          if (nn >= p.n) continue;
          uint16_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = accf[mt][4 * g + e];
            o[e] = from_f32<DT>(p.direct_bias ? v + to_f32<DT>(p.direct_bias[nn + e]) : v);
          }
          *reinterpret_cast<uint2*>(p.direct_out + (int64_t)m * p.n + nn) =
              uint2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
        }
      }
      return;
    }
  }
  // partial [split][m][n]: D[n][m] -- lane = column m (nl), rows n = 8g + 4h + e
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + nl;
    if (m >= p.m) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t nn = ntile + wv * 32 + 8 * g + 4 * h;
      if (nn >= p.n) continue;  // N % 4 == 0 (checked on the host): a group of four is in or out as a whole
      const int64_t off = ((int64_t)blockIdx.y * p.m + m) * p.n + nn;
      if constexpr (I8A)
        *reinterpret_cast<i32x4*>((int32_t*)p.part + off) =
            i32x4{acci[mt][4 * g], acci[mt][4 * g + 1], acci[mt][4 * g + 2], acci[mt][4 * g + 3]};
      else
        *reinterpret_cast<f32x4*>((float*)p.part + off) =
            f32x4{accf[mt][4 * g], accf[mt][4 * g + 1], accf[mt][4 * g + 2], accf[mt][4 * g + 3]};
    }
  }
}

// out[m][n] = epilogue(sum_s part[s][m][n]); one thread per 4 outputs
template <bool I8A, int DT = LL_F16>
__global__ __launch_bounds__(256) void dense8_finish(uint16_t* __restrict__ out, const void* __restrict__ part, int splits,
                                                     int64_t m, int64_t n, const uint16_t* __restrict__ bias,
                                                     const float* __restrict__ a_scale, const float* __restrict__ w_scale,
                                                     int32_t* __restrict__ acc_out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= m * n) return;
  const int64_t row = i / n, col = i - row * n;
  float v[4];
  // eight partial planes per round, requested together (a one-load-per-iteration loop pays the memory latency S times);
  // the sums are still formed in split order
  if constexpr (I8A) {
    i32x4 a = {0, 0, 0, 0};
    for (int s0 = 0; s0 < splits; s0 += 8) {
      i32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        t[u] = *reinterpret_cast<const i32x4*>((const int32_t*)part + (int64_t)(s0 + u < splits ? s0 + u : s0) * m * n + i);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < splits) a += t[u];
    }
    if (acc_out) *reinterpret_cast<i32x4*>(acc_out + i) = a;
    const float as = a_scale[row];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((float)a[e] * as) * w_scale[col + e];  // (acc.f32 * a_scale[m]) * w_scale[n], w8a8.py:118-120
  } else {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < splits; s0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        t[u] = *reinterpret_cast<const f32x4*>((const float*)part + (int64_t)(s0 + u < splits ? s0 + u : s0) * m * n + i);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < splits) a += t[u];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e];
  }
  uint16_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = from_f32<DT>(bias ? v[e] + to_f32<DT>(bias[col + e]) : v[e]);
  *reinterpret_cast<uint2*>(out + i) = uint2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
}

int d8_splits(int64_t n, int64_t k, int kch = 128, int cap = 1 << 30) {
  const int tiles = (int)((n + 127) / 128), chunks = (int)(k / kch);
  int s = (768 + tiles - 1) / tiles;   // ~3 resident workgroups per CU ...
  if (s > chunks / 2) s = chunks / 2;  // ... of at least two chunks each (the tile prefetch needs something to hide behind)
  if (s > cap) s = cap;                // (partial mode: what the consumer of the planes adds up in one pass)
  if (s < 1) s = 1;
  return s;
}

}  // namespace

// shapes the skinny engine takes (the dispatchers of gemm_wq.hip ask; also sizes the partial planes in ll_gemm_workspace)
extern "C" int64_t ll_dense8_partial_words(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 4 || n % 4 != 0 || k < 128 || k % 128 != 0) return 0;
  return (int64_t)d8_splits(n, k) * m * n;
}

// wfmt: 1 fp8 e4m3 / 2 int8 (fp16 activations, block scales) / 3 int8 x int8 (per-token / per-channel scales).
// Returns 1 when the launch was issued, 0 when the shape is not served (nothing happened), < 0 on a launch error.
extern "C" int ll_dense8_try(void* out, const void* x, const void* w, const float* scales, const float* a_scale,
                             const void* bias, int32_t* acc_out, int64_t m, int64_t n, int64_t k, int group_n,
                             int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride, int64_t s_stride_n,
                             int64_t s_stride_k, void* partials, void* stream) {
  static const bool off = getenv("LL_DENSE8_OFF") != nullptr;  // A/B knob, read once
  if (off || !partials || ll_dense8_partial_words(m, n, k) == 0) return 0;
  if (w_stride % 16 != 0 || !ll_aligned16(w) || !ll_aligned16(x) || !ll_aligned16(out) || !ll_aligned16(partials)) return 0;
  if (wfmt == D8_I8I8 ? (x_stride % 16 != 0) : (x_stride % 8 != 0 || (group_k < k && group_k % 64 != 0))) return 0;
  D8Params p{};
  p.part = partials; p.x = x; p.w = (const unsigned char*)w; p.scales = scales;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride; p.w_stride = w_stride;
  p.s_stride_n = s_stride_n; p.s_stride_k = s_stride_k; p.group_n = group_n > 0 ? group_n : 1;
  p.group_k = group_k > 0 ? group_k : k;
  p.chunks = (int)(k / 128);
  p.nt_w = d8_nt(n * k);
  const int splits = d8_splits(n, k);
  p.cps = (p.chunks + splits - 1) / splits;
  const int used = (p.chunks + p.cps - 1) / p.cps;  // every launched split has at least one chunk
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)used);
  hipStream_t st = (hipStream_t)stream;
#define LL_D8(WF)                                                        \
  if (m > 32) dense8_kernel<WF, 2><<<grid, 256, 0, st>>>(p);             \
  else dense8_kernel<WF, 1><<<grid, 256, 0, st>>>(p)
  if (wfmt == D8_FP8) { LL_D8(D8_FP8); }
  else if (wfmt == D8_I8) { LL_D8(D8_I8); }
  else if (wfmt == D8_I8I8) { LL_D8(D8_I8I8); }
  else return 0;
#undef LL_D8
  const int64_t quads = m * n / 4;
  const dim3 fgrid((unsigned)((quads + 255) / 256));
  if (wfmt == D8_I8I8)
    dense8_finish<true><<<fgrid, 256, 0, st>>>((uint16_t*)out, partials, used, m, n, (const uint16_t*)bias, a_scale, scales,
                                               acc_out);
  else
    dense8_finish<false><<<fgrid, 256, 0, st>>>((uint16_t*)out, partials, used, m, n, (const uint16_t*)bias, nullptr,
                                                nullptr, nullptr);
  return hipGetLastError() == hipSuccess ? 1 : LL_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------------------- //
// Unquantised 16-bit linears at decode shapes (round 3; SURVEY 8 row g1 / review "missing" 4): the reference runs these --
// the whole weight stream of an fp16 / bf16 model, the lm_head of every model, the MoE router -- through torch's F.linear
// (models/quantization/methods/unquantized.py:21-22, models/base.py:486-489), i.e. a vendor GEMM.  Here they take the same
// split-K weight-streaming kernel as the 8-bit formats with the dequantisation removed: out = x @ w^T (+ bias), fp32
// accumulation, one rounding to the activation dtype (the arithmetic of an fp16 / bf16 GEMM with fp32 accumulate; the
// summation order differs from the library's, like any two GEMM implementations).  x, w, bias, out: fp16 or bf16 (one dtype).
// ---------------------------------------------------------------------------------------------------------------- //
extern "C" int64_t ll_dense16_partial_words(int64_t m, int64_t n, int64_t k) {
  if (m < 1 || m > 64 || n < 4 || n % 4 != 0 || k < 64 || k % 64 != 0) return 0;
  return (int64_t)d8_splits(n, k, 64) * m * n;
}

// Returns 1 when the launches were issued, 0 when the shape / alignment is not served (nothing happened), < 0 on error.
extern "C" int ll_dense16_matmul(void* out, const void* x, const void* w, const void* bias, int64_t m, int64_t n, int64_t k,
                                 int64_t x_stride, int64_t w_stride, int dtype, void* partials, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (m == 0) return 1;
  if (!out || !x || !w || !partials || ll_dense16_partial_words(m, n, k) == 0) return 0;
  if (w_stride % 8 != 0 || x_stride % 8 != 0 || !ll_aligned16(w) || !ll_aligned16(x) || !ll_aligned16(out) || !ll_aligned16(partials))
    return 0;
  D8Params p{};
  p.part = partials; p.x = x; p.w = (const unsigned char*)w; p.scales = nullptr;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride; p.w_stride = w_stride * 2;  // bytes
  p.group_n = 1; p.group_k = k;
  p.chunks = (int)(k / 64);
  p.nt_w = d8_nt(n * k * 2);
  const int splits = d8_splits(n, k, 64);
  p.cps = (p.chunks + splits - 1) / splits;
  const int used = (p.chunks + p.cps - 1) / p.cps;
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)used);
  hipStream_t st = (hipStream_t)stream;
  if (used == 1) {
    p.direct_out = (uint16_t*)out;
    p.direct_bias = (const uint16_t*)bias;
  }
  if (dtype == LL_F16) {
    if (m > 32) dense8_kernel<D8_F16, 2><<<grid, 256, 0, st>>>(p);
    else dense8_kernel<D8_F16, 1><<<grid, 256, 0, st>>>(p);
  } else {
    if (m > 32) dense8_kernel<D8_BF16, 2><<<grid, 256, 0, st>>>(p);
    else dense8_kernel<D8_BF16, 1><<<grid, 256, 0, st>>>(p);
  }
  if (used == 1) return hipGetLastError() == hipSuccess ? 1 : LL_ERR_LAUNCH;
  const int64_t quads = m * n / 4;
  const dim3 fgrid((unsigned)((quads + 255) / 256));
  if (dtype == LL_F16)
    dense8_finish<false, LL_F16><<<fgrid, 256, 0, st>>>((uint16_t*)out, partials, used, m, n, (const uint16_t*)bias, nullptr,
                                                       nullptr, nullptr);
  else
    dense8_finish<false, LL_BF16><<<fgrid, 256, 0, st>>>((uint16_t*)out, partials, used, m, n, (const uint16_t*)bias, nullptr,
                                                        nullptr, nullptr);
  return hipGetLastError() == hipSuccess ? 1 : LL_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------------------- //
// Split-K partial mode of the skinny engine (round 4; the int4 engine's epilogue 2 for the other formats): the launch
// leaves its fp32 partial planes [S][M][N] -- block scales already applied (a9) / plain products (16-bit) -- for the
// consumer of the projection to add up: ll_skip_rmsnorm_partials after a row-parallel projection, ll_decode_attention_partials
// after the fused q|k|v one.  No dense8_finish launch, no second pass over the output.  wfmt: 1 fp8 e4m3, 2 int8 (fp16
// activations, block scales), 4 fp16, 5 bf16 weights (activations of the same type); 3 = a10 (int8 activations, quantised by
// the caller, x int8 weights): the planes are then EXACT int32 sums that still need a_scale[m] * w_scale[n] -- consumers:
// ll_skip_rmsnorm_q8, ll_w8a8_finish_swiglu (w8a8_fused.hip).  Returns the number of planes written (>= 1), 0 when the
// shape is not served, < 0 on error.
// ---------------------------------------------------------------------------------------------------------------- //
// the short-stream form of this mode (gemm_short_dense.hip): launches of a few tens of KB per CU
int sd_partials_slices(int64_t m, int64_t n, int64_t k, int wfmt, int max_splits);
int sd_launch(float* out, const void* x, const void* w, const float* scales, int64_t m, int64_t n, int64_t k, int group_n,
              int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride_bytes, int64_t s_stride_n, int64_t s_stride_k,
              int max_splits, void* stream);

extern "C" int ll_dense_partials_count(int64_t m, int64_t n, int64_t k, int wfmt, int max_splits) {
  const bool w16 = wfmt == D8_F16 || wfmt == D8_BF16;
  if (wfmt != D8_FP8 && wfmt != D8_I8 && wfmt != D8_I8I8 && !w16) return 0;
  if (const int s = sd_partials_slices(m, n, k, wfmt, max_splits)) return s;
  const int kch = w16 ? 64 : 128;
  if (m < 1 || m > 64 || n < 4 || n % 4 != 0 || k < kch || k % kch != 0 || max_splits < 1) return 0;
  const int chunks = (int)(k / kch);
  const int splits = d8_splits(n, k, kch, max_splits);
  const int cps = (chunks + splits - 1) / splits;
  return (chunks + cps - 1) / cps;
}

extern "C" int ll_dense_partials(float* partials, const void* x, const void* w, const float* scales, int64_t m, int64_t n,
                                 int64_t k, int group_n, int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride,
                                 int64_t s_stride_n, int64_t s_stride_k, int max_splits, void* stream) {
  const bool w16 = wfmt == D8_F16 || wfmt == D8_BF16;
  const int used = ll_dense_partials_count(m, n, k, wfmt, max_splits);
  if (used == 0) return 0;
  const bool i8a = wfmt == D8_I8I8;
  if (!partials || !x || !w || (!w16 && !i8a && !scales)) return LL_ERR_ARG;
  const int eb = w16 ? 2 : 1;
  if (sd_partials_slices(m, n, k, wfmt, max_splits))  // (declines -- 0 -- exactly where the streaming engine below would)
    return sd_launch(partials, x, w, scales, m, n, k, group_n, group_k, wfmt, x_stride, w_stride * eb, s_stride_n, s_stride_k,
                     max_splits, stream);
  if ((w_stride * eb) % 16 != 0 || x_stride % (i8a ? 16 : 8) != 0 || !ll_aligned16(w) || !ll_aligned16(x) || !ll_aligned16(partials))
    return 0;
  if (!w16 && !i8a && group_k < k && group_k % 64 != 0) return 0;
  D8Params p{};
  p.part = partials; p.x = x; p.w = (const unsigned char*)w; p.scales = scales;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride; p.w_stride = w_stride * eb;
  p.s_stride_n = s_stride_n; p.s_stride_k = s_stride_k; p.group_n = group_n > 0 ? group_n : 1;
  p.group_k = group_k > 0 ? group_k : k;
  const int kch = w16 ? 64 : 128;
  p.chunks = (int)(k / kch);
  p.nt_w = d8_nt(n * k * eb);
  p.cps = (p.chunks + used - 1) / used;
  if ((p.chunks + p.cps - 1) / p.cps != used) return LL_ERR_SHAPE;  // (cannot happen: `used` is a fixed point of the split rule)
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)used);
  hipStream_t st = (hipStream_t)stream;
#define LL_D8P(WF)                                                       \
  if (m > 32) dense8_kernel<WF, 2><<<grid, 256, 0, st>>>(p);             \
  else dense8_kernel<WF, 1><<<grid, 256, 0, st>>>(p)
  if (wfmt == D8_FP8) { LL_D8P(D8_FP8); }
  else if (wfmt == D8_I8) { LL_D8P(D8_I8); }
  else if (wfmt == D8_I8I8) { LL_D8P(D8_I8I8); }
  else if (wfmt == D8_F16) { LL_D8P(D8_F16); }
  else { LL_D8P(D8_BF16); }
#undef LL_D8P
  return hipGetLastError() == hipSuccess ? used : LL_ERR_LAUNCH;
}
