// M-tiled GEMMs over 8-bit weights for MANY rows (prefill: M = batch x prompt tokens) -- round 6.  Semantics:
//   a9  w8a16_matmul   lite_llama/kernels/quantization/w8a16.py:155-216: fp8-e4m3 / int8 weights [N, K], one fp32 scale per
//                      (group_n x group_k) block, fp16 activations, fp32 accumulation, + bias, one rounding to fp16;
//   a10 smoothquant    lite_llama/kernels/quantization/w8a8.py:151-217: int8 activations (per-token scale) x int8 weights
//                      (per-channel scale), EXACT int32 accumulation, (acc.f32 * a_scale[m]) * w_scale[n] + bias -> fp16.
// The reference's Triton kernels tile M x N; until this round every call of more than 64 rows looped the decode engines' 64-row
// weight-streaming tile over M (gemm_wq.hip::wgemm_kernel): a 16 384-row prompt re-streamed the weights 256 times.  Here, the
// shape of gemm_w4_prefill.hip:
//   * tile 256 token rows x 256 weight rows x (128 k int8 x int8 | 64 k for fp16 activations); 8 waves = 2 (M halves) x 4 (N
//     quarters), wave tile 128 x 64 = 8 MFMA 32x32 accumulators; A operand = weights, B operand = activations, so a lane ends up
//     with consecutive output columns of ONE token row (16-byte stores after one v_permlane32_swap);
//   * both tiles go global -> LDS by LDS-DMA as whole 128-byte (64-byte: fp8 / int8 weights under fp16 activations) row pieces,
//     XOR-swizzled through the per-lane SOURCE address (slot s of row r at s ^ ((r >> 1) & 7)): 16-byte fragment reads are
//     bank-conflict free; double-buffered, one barrier per k-step, the next step's tiles in flight under this step's 32 MFMAs;
//   * int8 x int8 on mfma_i32_32x32x32_i8 (exact); fp8 / int8 weights widened in registers by the decode engines' exact bit
//     surgery (gemm_w8_common.h, one definition) with the block scale folded in, then mfma_f32_32x32x16_f16.
#include <stdlib.h>

#include "common.h"
#include "gemm_w4_common.h"
#include "gemm_w8_common.h"

#define P8_BN 256
#ifndef P8_WMH_DEFAULT
#define P8_WMH_DEFAULT 1  // 128-row M halves per workgroup: 1 = 128 x 256 tiles on 4 waves, two workgroups per CU (round 6; see gemm_w4_prefill.hip)
#endif

struct P8Params {
  uint16_t* out;            // fp16 [M][N]
  const unsigned char* x;   // fp16 [M][K] (a9) or int8 [M][K] (a10); row stride x_stride BYTES
  const unsigned char* w;   // [N][K] bytes, row stride w_stride bytes
  const float* scales;      // a9: block scales; a10: w_scale [N]
  const float* a_scale;     // a10: [M]
  const uint16_t* bias;     // fp16 [N] or nullptr
  int32_t* acc_out;         // a10: optional raw int32 sums [M][N]
  int64_t m, n, k, x_stride, w_stride, s_stride_n, s_stride_k, group_k;
  int group_n;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int p8_swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// WF: 1 fp8 e4m3 weights, 2 int8 weights (fp16 activations; block scales), 3 int8 x int8 (per-token x per-channel scales)
// WMH: 128-row M halves per workgroup (2: the 256 x 256 tile on 8 waves, one workgroup per CU; 1: 128 x 256 on 4 waves, two per CU --
// one's per-k-step barrier stall is covered by the other's MFMAs)
template <int WF, int WMH>
__global__ __launch_bounds__(WMH * 256) __attribute__((amdgpu_waves_per_eu(2, 2))) void w8_mtiled_kernel(const P8Params p) {
  constexpr bool I8A = WF == 3;
  constexpr int P8_BM = WMH * 128;
  constexpr int BK = I8A ? 128 : 64;            // k per step
  constexpr int XT = P8_BM * 128;               // activation tile: P8_BM rows x 128 B
  constexpr int WROW = I8A ? 128 : 64;          // bytes of a weight row per step
  constexpr int WT = 256 * WROW;
  constexpr int BUF = XT + WT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  {  // block -> tile: groups of 16 x 16 tiles; workgroup 8 j + xcd takes an 8 (m) x 4 (n) corner of the group (gemm_w4_prefill.hip)
    const int b = (int)blockIdx.x;
    const int g = b >> 8, i = b & 255, xcd = i & 7, j = i >> 3;
    const int groups_m = (p.tiles_m + 15) >> 4;
    const int gm = g % groups_m, gn = g / groups_m;
    tm = gm * 16 + (xcd & 1) * 8 + (j & 7);
    tn = gn * 16 + (xcd >> 1) * 4 + (j >> 3);
  }
  if (tm >= p.tiles_m || tn >= p.tiles_n) return;
  const int64_t m0 = (int64_t)tm * P8_BM, n0 = (int64_t)tn * P8_BN;
  const int ksteps = (int)(p.k / BK);

  // ---- loader role of every wave: activation pieces 4 w .. 4 w + 3 (8 rows x 128 B each); weight pieces (8 rows x 128 B, or
  //      16 rows x 64 B) 4 w .. / 2 w ..; lane -> (row, physical 16-byte slot) ----
  uint32_t xoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wv * 4 + j) * 8 + (lane >> 3), ps = lane & 7;
    int64_t row = m0 + r;
    if (row >= p.m) row = p.m - 1;  // rows >= M feed only unstored outputs
    xoff[j] = (uint32_t)(row * p.x_stride + p8_swz(r, ps) * 16);
  }
  constexpr int WP = (I8A ? 4 : 2) * (2 / WMH);  // weight pieces (1 KB) per wave and step (the 256-row weight tile over 4 or 8 waves)
  uint32_t woff[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    int r, src_slot;
    if constexpr (I8A) {
      r = (wv * WP + j) * 8 + (lane >> 3);
      src_slot = p8_swz(r, lane & 7);
    } else {
      // LDS image: [128 row pairs][128 B]; physical slot ps of pair pr holds (row 2 pr + (ls >> 2), 16-byte piece ls & 3) with
      // ls = ps ^ ((pr >> 1) & 7): an 8-byte fragment read of 32 consecutive rows touches every bank pair at most twice
      const int pr = (wv * WP + j) * 8 + (lane >> 3), ls = p8_swz(pr, lane & 7);
      r = pr * 2 + (ls >> 2);
      src_slot = ls & 3;
    }
    int64_t row = n0 + r;
    if (row >= p.n) row = p.n - 1;  // rows >= N feed only unstored outputs
    woff[j] = (uint32_t)(row * p.w_stride + src_slot * 16);
  }
  auto issue = [&](int ks, int buf) {
    const unsigned char* xb = p.x + (size_t)ks * 128;  // 128 B of k per step either way (128 int8 / 64 fp16)
    const unsigned char* wb = p.w + (size_t)ks * WROW;
#pragma unroll
    for (int j = 0; j < 4; ++j) v3_dma16<false>((uint32_t)(buf * BUF + (wv * 4 + j) * 1024), xb, xoff[j]);
#pragma unroll
    for (int j = 0; j < WP; ++j) v3_dma16<false>((uint32_t)(buf * BUF + XT + (wv * WP + j) * 1024), wb, woff[j]);
  };

  // ---- consumer geometry: wave (wm, wn) multiplies token rows wm * 128 .. + 127 by weight rows wn * 64 .. + 63 ----
  const int wm = WMH == 2 ? wv >> 2 : 0, wn = wv & 3;
  const int nl = lane & 31, h = lane >> 5;
  int xrow_off[4], xsw[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int r = wm * 128 + mi * 32 + nl;
    xrow_off[mi] = r * 128;
    xsw[mi] = (r >> 1) & 7;
  }
  int wrow_off[2], wsw[2], wsub[2];
  const float* srow[2] = {nullptr, nullptr};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int r = wn * 64 + ni * 32 + nl;
    if constexpr (I8A) {
      wrow_off[ni] = r * 128;
      wsw[ni] = (r >> 1) & 7;
      wsub[ni] = 0;
    } else {
      wrow_off[ni] = (r >> 1) * 128;
      wsw[ni] = ((r >> 1) >> 1) & 7;
      wsub[ni] = (r & 1) * 4;
      int64_t row = n0 + r;
      if (row >= p.n) row = p.n - 1;
      srow[ni] = p.scales + (row / p.group_n) * p.s_stride_n;
    }
  }
  f32x16 accf[2][4];
  i32x16 acci[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        accf[ni][mi][e] = 0.f;
        acci[ni][mi][e] = 0;
      }
  auto scale_of = [&](int ks, int ni) -> uint32_t {  // the (row, k-group) block scale as a broadcast fp16 pair (x 256 for fp8)
    const float sv = srow[ni][(int64_t)(((int64_t)ks * BK) / p.group_k) * p.s_stride_k];
    return d8_bcast(WF == 1 ? sv * 256.0f : sv);
  };

  issue(0, 0);
  uint32_t sp_next[2] = {0, 0};
  if constexpr (!I8A) {
    sp_next[0] = scale_of(0, 0);
    sp_next[1] = scale_of(0, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int ks = 0; ks < ksteps; ++ks) {
    const unsigned char* xb = lds + (ks & 1) * BUF;
    const unsigned char* wb = xb + XT;
    uint32_t sp[2] = {sp_next[0], sp_next[1]};
    if (ks + 1 < ksteps) {
      issue(ks + 1, (ks + 1) & 1);  // that buffer was read in step ks - 1
      if constexpr (!I8A) {
        sp_next[0] = scale_of(ks + 1, 0);
        sp_next[1] = scale_of(ks + 1, 1);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (I8A) {
        i32x4 xf[4], wf[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[mi] = *reinterpret_cast<const i32x4*>(xb + xrow_off[mi] + (((kk * 2 + h) ^ xsw[mi]) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) wf[ni] = *reinterpret_cast<const i32x4*>(wb + wrow_off[ni] + (((kk * 2 + h) ^ wsw[ni]) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acci[ni][mi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ni], xf[mi], acci[ni][mi], 0, 0, 0);
      } else {
        f16x8 xf[4], wf[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[mi] = *reinterpret_cast<const f16x8*>(xb + xrow_off[mi] + (((kk * 2 + h) ^ xsw[mi]) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          // MFMA step kk contracts k = 16 kk + 8 h .. + 7: 8 weight bytes = half h of the row's 16-byte piece kk
          const uint2 raw = *reinterpret_cast<const uint2*>(wb + wrow_off[ni] + (((wsub[ni] + kk) ^ wsw[ni]) * 16) + h * 8);
          uint32_t d0, d1, d2, d3;
          if constexpr (WF == 1) {
            d8_fp8(raw.x, sp[ni], d0, d1);
            d8_fp8(raw.y, sp[ni], d2, d3);
          } else {
            d8_i8(raw.x, sp[ni], d0, d1);
            d8_i8(raw.y, sp[ni], d2, d3);
          }
          wf[ni] = __builtin_bit_cast(f16x8, u32x4{d0, d1, d2, d3});
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) accf[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], xf[mi], accf[ni][mi], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: a lane holds, for token row m = nl, the weight rows 8 g + 4 h .. + 3 of a 32-row group ----
  auto swap32 = [](uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  };
  const bool has_bias = p.bias != nullptr;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t mrow = m0 + wm * 128 + mi * 32 + nl;
    const bool row_ok = mrow < p.m;
    float as = 1.f;
    if constexpr (I8A) as = p.a_scale[row_ok ? mrow : 0];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t ncol = n0 + wn * 64 + ni * 32;
      if (ncol >= p.n) continue;  // (n % 32 == 0: a 32-row group is in or out as a whole)
      float v[16];
      if constexpr (I8A) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(p.scales + ncol + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * g + e] = ((float)acci[ni][mi][4 * g + e] * as) * ws[e];  // w8a8.py:118-120
          if (p.acc_out && row_ok)
            *reinterpret_cast<i32x4*>(p.acc_out + mrow * p.n + ncol + 8 * g + 4 * h) =
                i32x4{acci[ni][mi][4 * g], acci[ni][mi][4 * g + 1], acci[ni][mi][4 * g + 2], acci[ni][mi][4 * g + 3]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = accf[ni][mi][e];
      }
      if (has_bias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ncol + 8 * g + 4 * h);
          v[4 * g + 0] += f16_bits_to_f32((uint16_t)(bb.x & 0xffffu));
          v[4 * g + 1] += f16_bits_to_f32((uint16_t)(bb.x >> 16));
          v[4 * g + 2] += f16_bits_to_f32((uint16_t)(bb.y & 0xffffu));
          v[4 * g + 3] += f16_bits_to_f32((uint16_t)(bb.y >> 16));
        }
      }
      uint32_t lo[4], hi[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f = v[4 * g + e];
          asm volatile("" : "+v"(f));  // the fp32 value is rounded ONCE, here (never a fused multiply-convert: see w8a8_fused.hip)
          o[e] = f32_to_f16_bits(f);
        }
        lo[g] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        hi[g] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
      }
      const int64_t nn = ncol + 16 * h;
      swap32(lo[0], lo[2]);
      swap32(hi[0], hi[2]);
      swap32(lo[1], lo[3]);
      swap32(hi[1], hi[3]);
      if (row_ok) {
        *reinterpret_cast<u32x4*>(p.out + mrow * p.n + nn) = u32x4{lo[0], hi[0], lo[2], hi[2]};
        *reinterpret_cast<u32x4*>(p.out + mrow * p.n + nn + 8) = u32x4{lo[1], hi[1], lo[3], hi[3]};
      }
    }
  }
}

// wfmt: 1 fp8 e4m3 / 2 int8 (fp16 activations, block scales), 3 int8 x int8.  Shapes the M-tiled engine takes (the dispatchers of
// gemm_wq.hip ask): more than 64 rows (the decode engines stream the weights once for fewer), n % 32 == 0, k a multiple of the
// k-step, scale groups along k whole multiples of the step.
extern "C" int ll_w8_mtiled_supported(int64_t m, int64_t n, int64_t k, int wfmt, int64_t group_k) {
  static const bool off = getenv("LL_W8_NO_MTILED") != nullptr;  // A/B knob, read once
  if (off || (wfmt != 1 && wfmt != 2 && wfmt != 3)) return 0;
  const int bk = wfmt == 3 ? 128 : 64;
  if (m <= 64 || n < 32 || n % 32 != 0 || k < bk || k % bk != 0) return 0;
  if (wfmt != 3 && group_k < k && group_k % bk != 0) return 0;
  if (n * k >= (1ll << 31) || m * k * (wfmt == 3 ? 1 : 2) >= (1ll << 32)) return 0;  // 32-bit offsets into the operands
  const int64_t tiles = ((m + 127) / 128 + 15) / 16 * 16 * (((n + P8_BN - 1) / P8_BN + 15) / 16 * 16);
  return tiles < (1ll << 31) ? 1 : 0;
}

// Returns 1 when the launch was issued, 0 when the shape / alignment is not served (nothing happened), < 0 on a launch error.
extern "C" int ll_w8_mtiled_try(void* out, const void* x, const void* w, const float* scales, const float* a_scale, const void* bias,
                                int32_t* acc_out, int64_t m, int64_t n, int64_t k, int group_n, int64_t group_k, int wfmt,
                                int64_t x_stride, int64_t w_stride, int64_t s_stride_n, int64_t s_stride_k, void* stream) {
  if (group_k <= 0 || group_k > k) group_k = k;
  if (!ll_w8_mtiled_supported(m, n, k, wfmt, group_k)) return 0;
  const int eb = wfmt == 3 ? 1 : 2;  // bytes per activation element
  if ((x_stride * eb) % 16 != 0 || w_stride % 16 != 0 || !ll_aligned16(x) || !ll_aligned16(w) || !ll_aligned16(out)) return 0;
  if ((m - 1) * x_stride * eb + k * eb >= (1ll << 32) || (n - 1) * w_stride + k >= (1ll << 32)) return 0;
  if (wfmt == 3 && (!a_scale || !scales || !ll_aligned16(scales) || (acc_out && !ll_aligned16(acc_out)))) return 0;
  if (wfmt != 3 && !scales) return 0;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) return 0;
  P8Params p{};
  p.out = (uint16_t*)out; p.x = (const unsigned char*)x; p.w = (const unsigned char*)w; p.scales = scales; p.a_scale = a_scale;
  p.bias = (const uint16_t*)bias; p.acc_out = acc_out;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride * eb; p.w_stride = w_stride;
  p.s_stride_n = s_stride_n; p.s_stride_k = s_stride_k; p.group_k = group_k; p.group_n = group_n > 0 ? group_n : 1;
  static const int wmh_env = getenv("LL_P8_WMH") ? atoi(getenv("LL_P8_WMH")) : 0;  // A/B knob, read once (1 / 2: force the tile height)
  // measured (benchmarks/prefill_gemm8.py, round 6): the 128-row tile (two workgroups per CU) wins where the widening VALU sits in front
  // of the MFMAs (fp8 / int8 -> fp16: 487 -> 545, 393 -> 463 TFLOP/s) and loses on int8 x int8, whose MFMAs run twice as fast and leave
  // the L2 -> LDS tile traffic as the bound (2.0 -> 1.6 POP/s): that form keeps the 256-row tile
  const int wmh = wmh_env == 1 || wmh_env == 2 ? wmh_env : (wfmt == 3 ? 2 : P8_WMH_DEFAULT);
  const int bm = wmh * 128;
  p.tiles_m = (int)((m + bm - 1) / bm);
  p.tiles_n = (int)((n + P8_BN - 1) / P8_BN);
  const int64_t grid = (int64_t)((p.tiles_m + 15) / 16) * ((p.tiles_n + 15) / 16) * 256;
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  (void)hipGetDevice(&dev);
#define P8_GO(WF, WH, WROWB)                                                                                              \
  {                                                                                                                       \
    const int bytes_ = 2 * (WH * 128 * 128 + 256 * WROWB);                                                                \
    static bool attr_[16] = {false};                                                                                      \
    if (dev >= 0 && dev < 16 && !attr_[dev]) {                                                                            \
      (void)hipFuncSetAttribute((const void*)w8_mtiled_kernel<WF, WH>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_); \
      attr_[dev] = true;                                                                                                  \
    }                                                                                                                     \
    w8_mtiled_kernel<WF, WH><<<dim3((unsigned)grid), WH * 256, bytes_, st>>>(p);                                          \
  }
  if (wmh == 2) {
    if (wfmt == 1) P8_GO(1, 2, 64)
    else if (wfmt == 2) P8_GO(2, 2, 64)
    else P8_GO(3, 2, 128)
  } else {
    if (wfmt == 1) P8_GO(1, 1, 64)
    else if (wfmt == 2) P8_GO(2, 1, 64)
    else P8_GO(3, 1, 128)
  }
#undef P8_GO
  return hipGetLastError() == hipSuccess ? 1 : LL_ERR_LAUNCH;
}
