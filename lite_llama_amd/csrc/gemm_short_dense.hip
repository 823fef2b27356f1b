// SHORT-STREAM decode GEMM for the ROW-MAJOR weight formats (round 6): split-K partial launches of a few tens of KB per CU over
// fp8-e4m3 / int8 weights with block scales (a9, lite_llama/kernels/quantization/w8a16.py:155-216) and unquantised fp16 / bf16
// weights (the reference's F.linear, methods/unquantized.py:21-22) -- the dense q|k|v / o projections of Qwen3-30B-A3B (fp8, 10.5 /
// 8.4 MB), the narrow projections of a 1.5B bf16 model, every projection of a TP >= 4 shard.  Same shape as gemm_short.hip (the
// int4 engine): one workgroup per CU in ONE round, work item = (R row groups of 32 weight rows, one k-slice); the 8 consumer waves
// request ALL their weight bytes in their first instructions, straight into registers (a piece = 32 rows x 64 k = 2 / 4 KB: lane
// (row nl, half h) holds the row's k = 32 h .. 32 h + 31, i.e. the four MFMA k-steps' 8-value fragments in order -- whole 64 /
// 128-byte runs of a row per lane pair, non-temporal); 4 loader waves bring the k-slice of the activations into LDS once (two
// stages); a piece is widened (the decode engines' exact bit surgery, gemm_w8_common.h, block scale folded in) and multiplied as it
// lands; the k-interleaves of a row group meet through LDS in wave order and leave as whole 128-byte lines of the fp32 plane
// [slice][m][n] -- the planes ll_dense_partials documents, for ll_skip_rmsnorm_partials / ll_decode_attention_partials.
// The launches this plan declines (long streams, shapes off its grid) stay on gemm_w8_skinny.hip.
#include <stdlib.h>

#include "common.h"
#include "gemm_w4_common.h"  // u32x4 / u32x2, v3_dma16, v3_vmcnt
#include "gemm_w8_common.h"

#define SD_CONSUMERS 8
#define SD_LOADERS 4
#define SD_THREADS ((SD_CONSUMERS + SD_LOADERS) * 64)
#define SD_MAX_CHUNKS 9  // activation chunks (128 k) of one k-slice resident in LDS
#define SD_MAX_SLICES 8

typedef __bf16 sd_bf16x8 __attribute__((ext_vector_type(8)));

struct SDParams {
  const char* w;       // [N][K] weights, row stride w_stride BYTES
  const float* sc;     // 8-bit: one fp32 per (group_n rows, gkb 64-k blocks); nullptr for 16-bit weights
  int n, w_stride, group_n, gkb, s_stride_n, s_stride_k;
  int kb_base, kb_rem; // k-slice s owns the 64-k blocks [s kb_base + min(s, kb_rem), + kb_base + (s < kb_rem))
  int RB;              // row blocks (R row groups each)
  int total, cap;      // work items = S * RB, item = slice * RB + row block; workgroup b takes item (b % 8) cap + b / 8
  int m;
  const uint16_t* x;   // [M][K] fp16 (bf16 for bf16 weights), row stride x_stride elements
  float* out;          // [S][M][N] fp32
  int x_stride;
};

__device__ __forceinline__ void sd_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void sd_wait_outstanding(int n) {  // at most n of this wave's memory operations still in flight
  switch (n) {
    case 0: v3_vmcnt<0>(); break;
    case 2: v3_vmcnt<2>(); break;
    case 4: v3_vmcnt<4>(); break;
    case 6: v3_vmcnt<6>(); break;
    case 8: v3_vmcnt<8>(); break;
    case 12: v3_vmcnt<12>(); break;
    case 16: v3_vmcnt<16>(); break;
    default: v3_vmcnt<0>(); break;
  }
}

// WF: 1 fp8 e4m3, 2 int8 (fp16 activations, block scales), 4 fp16, 5 bf16 weights (activations of the weight's type).
// MT: 32-row batch halves; R: row groups per work item (8 / R k-interleaves per row group); P: pieces per consumer wave at most.
template <int WF, int MT, int R, int P>
__global__ __launch_bounds__(SD_THREADS) void wssd_kernel(const SDParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int KQ = SD_CONSUMERS / R;
  constexpr int EW = (WF == 4 || WF == 5) ? 2 : 1;  // bytes per weight
  constexpr int NW = 2 * EW;                        // 16-byte registers of a piece per lane (32 weights)
  constexpr int PH = P / 2;                         // pieces per activation stage
  constexpr int XT = MT * 32 * 256;                 // one chunk's activation tile: [MT * 32 rows][16 x 16 B], slot j of row r at j ^ (r & 15)
  static_assert(P % 2 == 0, "two activation stages");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = ((int)blockIdx.x & 7) * p.cap + ((int)blockIdx.x >> 3);
  if (item >= p.total) return;
  const int s = item / p.RB, rb = item - s * p.RB;
  const int kb_lo = s * p.kb_base + (s < p.kb_rem ? s : p.kb_rem);
  const int nkb = p.kb_base + (s < p.kb_rem ? 1 : 0);
  const int c_lo = kb_lo >> 1;
  const int nC = ((kb_lo + nkb + 1) >> 1) - c_lo;  // activation chunks the slice touches
  int nC0 = ((kb_lo + PH * KQ - 1) >> 1) - c_lo + 1;  // stage 0 (barrier A): the chunks of every consumer's first PH pieces
  if (nC0 > nC) nC0 = nC;

  if (wv >= SD_CONSUMERS) {
    // ============================== activation loaders ============================== //
    constexpr int PPL = MT * 8 / SD_LOADERS;  // 1-KB pieces (4 rows x 256 B) of a chunk tile per loader wave
    const int L = wv - SD_CONSUMERS;
    uint32_t voff[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      int row = (L * PPL + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (row & 15);
      if (row >= p.m) row = p.m - 1;  // rows >= M feed only unstored outputs
      voff[j] = (uint32_t)(row * p.x_stride * 2 + slot * 16);
    }
    const char* xb = (const char*)p.x + (size_t)c_lo * 256;
    for (int c = 0; c < nC; ++c) {
#pragma unroll
      for (int j = 0; j < PPL; ++j) v3_dma16<false>((uint32_t)(c * XT + (L * PPL + j) * 1024), xb + (size_t)c * 256, voff[j]);
    }
    sd_wait_outstanding((nC - nC0) * PPL);
    sd_barrier();  // A: chunks [0, nC0) of the slice are in LDS
    v3_vmcnt<0>();
    sd_barrier();  // B: all of it
    return;
  }

  // ================================= consumers ================================= //
  const int r = wv % R, q = wv / R;
  const int rg = rb * R + r;  // row group of 32 weight rows
  const int nl = lane & 31, h = lane >> 5;
  u32x4 w[P][NW];
  float sc[P];
  {
    const char* wrow = p.w + (size_t)(uint32_t)(rg * 32 + nl) * (uint32_t)p.w_stride + (uint32_t)(32 * h * EW);
    const float* srow = p.sc ? p.sc + (size_t)(uint32_t)((rg * 32 + nl) / p.group_n) * (uint32_t)p.s_stride_n : nullptr;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int kk = i * KQ + q;
      const int kb = kb_lo + (kk < nkb ? kk : 0);  // pieces past the slice re-read its first one (multiplied by zero)
#pragma unroll
      for (int u = 0; u < NW; ++u)
        w[i][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)(uint32_t)kb * (64 * EW) + u * 16));
      if constexpr (EW == 1) sc[i] = srow[(size_t)(uint32_t)(kb / p.gkb) * (uint32_t)p.s_stride_k];
      else sc[i] = 1.f;
    }
  }
  int x_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x_off[j] = nl * 256 + (((h * 4 + j) ^ (nl & 15)) * 16);
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

  auto read_x = [&](f16x8 (&a)[4][MT], int i) {
    const int kk = i * KQ + q;
    const int kb = kb_lo + (kk < nkb ? kk : 0);  // (a piece past the slice: finite, landed data x exact zeros)
    const unsigned char* xb = lds + ((kb >> 1) - c_lo) * XT;
    const int kx = (kb & 1) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[j][mt] = *reinterpret_cast<const f16x8*>(xb + (x_off[j] ^ kx) + mt * 32 * 256);
  };
  auto mul = [&](const f16x8 (&a)[4][MT], int i) {
    const bool live = (i * KQ + q) < nkb;
    uint32_t sp = 0u;
    if constexpr (EW == 1) sp = d8_bcast(live ? (WF == 1 ? sc[i] * 256.0f : sc[i]) : 0.f);  // a piece past the slice: scale 0
    const uint32_t keep = live ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x4 fr;
      if constexpr (EW == 1) {
        // MFMA step j contracts the lane's weights 8 j .. 8 j + 7 = bytes [8 j, 8 j + 8) of its 32
        const u32x4 raw = w[i][j >> 1];
        const uint32_t lo = (j & 1) ? raw.z : raw.x, hi = (j & 1) ? raw.w : raw.y;
        uint32_t d0, d1, d2, d3;
        if constexpr (WF == 1) {
          d8_fp8(lo, sp, d0, d1);
          d8_fp8(hi, sp, d2, d3);
        } else {
          d8_i8(lo, sp, d0, d1);
          d8_i8(hi, sp, d2, d3);
        }
        fr = u32x4{d0, d1, d2, d3};
      } else {
        fr = w[i][j];
        fr.x &= keep; fr.y &= keep; fr.z &= keep; fr.w &= keep;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (WF == 5)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sd_bf16x8, fr), __builtin_bit_cast(sd_bf16x8, a[j][mt]), acc[mt], 0, 0, 0);
        else
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr), a[j][mt], acc[mt], 0, 0, 0);
      }
    }
  };
  f16x8 a0[4][MT], a1[4][MT];
  sd_barrier();  // A
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    if (st == 1) sd_barrier();  // B
    read_x(a0, st * PH);
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      if (i + 1 < PH) read_x((i & 1) ? a0 : a1, st * PH + i + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads up here (hipcc sinks them to their use otherwise)
      mul((i & 1) ? a1 : a0, st * PH + i);
    }
  }

  // The KQ partial sums of a row group meet through the LDS the activation slice no longer needs: [wave][batch row][8 x 16 B],
  // 16-byte slot j of batch row m stored at j ^ ((m >> 1) & 7), summed in wave order, stored as whole 128-byte lines of the plane
  // (gemm_short.hip's exchange).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every consumer has read its last activation fragment
  {
    unsigned char* mine = lds + (size_t)wv * (MT * 32 * 128);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 32 + nl;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int quad = 2 * g + h;
        *reinterpret_cast<f32x4*>(mine + m * 128 + ((quad ^ ((m >> 1) & 7)) * 16)) =
            f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    constexpr int QW = MT * 256 / KQ;  // 16-byte quads of the row group's [MT * 32][32] fp32 output this wave finishes
    float* plane = p.out + (size_t)s * p.m * p.n + (size_t)rg * 32;
#pragma unroll
    for (int f0 = 0; f0 < QW; f0 += 64) {
      const int f = f0 + lane;
      if (QW >= 64 || f < QW) {
        const int F = q * QW + f;
        const int m = F >> 3, quad = F & 7;
        const unsigned char* src = lds + (size_t)(r * (MT * 32 * 128)) + m * 128 + ((quad ^ ((m >> 1) & 7)) * 16);
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int q2 = 1; q2 < KQ; ++q2) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(src + (size_t)q2 * R * (MT * 32 * 128));
          v += o;
        }
        if (m < p.m) *reinterpret_cast<f32x4*>(plane + (size_t)m * p.n + quad * 4) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct SDPlan {
  int ok = 0;
  int R = 0, S = 0, P = 0, kb_base = 0, kb_rem = 0, RB = 0, total = 0, cap = 0, grid = 0, lds = 0;
};

static int sd_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

static int sd_max_chunks(int kbs, int S) {  // most activation chunks any slice of the (kbs blocks, S slices) geometry touches
  const int base = kbs / S, rem = kbs % S;
  int worst = 0;
  for (int s = 0; s < S; ++s) {
    const int lo = s * base + (s < rem ? s : rem), n = base + (s < rem ? 1 : 0);
    const int nc = ((lo + n + 1) >> 1) - (lo >> 1);
    if (nc > worst) worst = nc;
  }
  return worst;
}

// (R, S) by gemm_short.hip's measured cost model with this format's bytes per piece; pieces per wave <= 8 (8-bit: 64 VGPRs) / 4
// (16-bit: 64 VGPRs); R in {1, 2, 4}.
static SDPlan sd_plan(int64_t m, int64_t n, int64_t k, int wfmt, int max_splits) {
  SDPlan best;
  static const bool off = getenv("LL_DENSE_SS") != nullptr && atoi(getenv("LL_DENSE_SS")) == 0;  // A/B knobs, read once
  static const int force_r = getenv("LL_DENSE_SS_R") ? atoi(getenv("LL_DENSE_SS_R")) : 0;
  static const int force_s = getenv("LL_DENSE_SS_S") ? atoi(getenv("LL_DENSE_SS_S")) : 0;
  const bool w16 = wfmt == 4 || wfmt == 5;
  if (off || (wfmt != 1 && wfmt != 2 && !w16) || m < 1 || m > 64 || n < 32 || n % 32 != 0 || k < 64 || k % 64 != 0) return best;
  if (n * k * (w16 ? 2 : 1) >= (1ll << 31) || (int64_t)(m - 1) * k * 2 + k * 2 >= (1ll << 31)) return best;  // 32-bit offsets
  const int cus = sd_num_cus();
  const int rgs = (int)(n / 32), kbs = (int)(k / 64);
  const int mt = m > 32 ? 2 : 1;
  const double piece_kb = w16 ? 4.0 : 2.0;
  const int pmax = w16 ? 4 : 8;
  double best_cost = 1e30;
  for (int R = 1; R <= 4; R *= 2) {
    if (rgs % R || (force_r && R != force_r)) continue;
    const int KQ = 8 / R, RB = rgs / R;
    for (int S = 1; S <= SD_MAX_SLICES && S <= max_splits; ++S) {
      if (S > kbs) break;
      if (force_s && S != force_s) continue;
      const int total = RB * S;
      if (total > cus) continue;
      const int nkb = (kbs + S - 1) / S;
      const int pw = (nkb + KQ - 1) / KQ;
      const int P = ((pw + 1) / 2) * 2;
      if (P > pmax) continue;
      const int mc = sd_max_chunks(kbs, S);
      if (mc > SD_MAX_CHUNKS) continue;
      const int slots = P * KQ;
      const double w_us = (double)R * nkb * piece_kb / 27.0;
      const double hbm = (double)n * (double)k * (w16 ? 2.0 : 1.0) / 6.5e6;
      const double x_us = (double)mc * mt * 8.0 / 100.0;
      const double p_us = (double)R * mt * 4.0 / 25.0;
      const double cost = (hbm > w_us ? hbm : w_us) + x_us + p_us + 0.15 * S + 0.02 * R * (slots - nkb);
      if (cost < best_cost) {
        best_cost = cost;
        best.ok = 1; best.R = R; best.S = S; best.P = P;
        best.kb_base = kbs / S; best.kb_rem = kbs % S; best.RB = RB; best.total = total;
        best.cap = (total + 7) / 8; best.grid = best.cap * 8;
        const int xbytes = mc * mt * 32 * 256, rbytes = 8 * mt * 32 * 128;
        best.lds = xbytes > rbytes ? xbytes : rbytes;
      }
    }
  }
  // a launch that would leave more than a third of the CUs idle stays on the streaming engine
  if (best.ok && best.total * 3 < cus * 2) best.ok = 0;
  // Measured (benchmarks/gemm_short_dense.py, same box, us per launch short-stream / streaming engine; DESIGN_NOTEBOOK.md 8.7):
  // fp8 2048 x 4096 7.4 / 11.2, int8 4096 x 1024 5.4 / 6.6 -- but fp8 5120 x 2048 (10.5 MB) 10.6 / 10.1 under every plan, and
  // the 16-bit formats 7.2 - 8.1 / 6.0 - 6.5: a lane's 32 - 64 contiguous bytes per row make every load instruction touch 32+ cache
  // lines (the int4 engine's pre-packed pieces are 1 KB contiguous).  Default: 8-bit weights of at most 9 MB; LL_DENSE_SS=2 lifts it.
  static const bool all = getenv("LL_DENSE_SS") != nullptr && atoi(getenv("LL_DENSE_SS")) == 2;
  if (best.ok && !all && (w16 || n * k > 9000000)) best.ok = 0;
  return best;
}

// planes the dense short-stream engine leaves for (m, n, k, format); 0: the launch is not its (gemm_w8_skinny.hip asks)
int sd_partials_slices(int64_t m, int64_t n, int64_t k, int wfmt, int max_splits) {
  const SDPlan pl = sd_plan(m, n, k, wfmt, max_splits);
  return pl.ok ? pl.S : 0;
}

template <int WF, int MT, int R, int P>
static void sd_go(const SDParams& p, const SDPlan& pl, hipStream_t st) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wssd_kernel<WF, MT, R, P>, hipFuncAttributeMaxDynamicSharedMemorySize, SD_MAX_CHUNKS * 2 * 32 * 256);
    attr_set[dev] = true;
  }
  wssd_kernel<WF, MT, R, P><<<dim3((unsigned)pl.grid), SD_THREADS, pl.lds, st>>>(p);
}

// Returns the planes written (the plan's S), 0 when the call is declined (alignment, scale grid), < 0 on a launch error.  Shapes
// were sized by sd_partials_slices.
int sd_launch(float* out, const void* x, const void* w, const float* scales, int64_t m, int64_t n, int64_t k, int group_n,
              int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride_bytes, int64_t s_stride_n, int64_t s_stride_k,
              int max_splits, void* stream) {
  const SDPlan pl = sd_plan(m, n, k, wfmt, max_splits);
  if (!pl.ok) return 0;
  const bool w16 = wfmt == 4 || wfmt == 5;
  if (w_stride_bytes % 16 != 0 || x_stride % 8 != 0 || !ll_aligned16(w) || !ll_aligned16(x) || !ll_aligned16(out)) return 0;
  if (w_stride_bytes >= (1ll << 31) || (int64_t)(n - 1) * w_stride_bytes + k * (w16 ? 2 : 1) >= (1ll << 32) ||
      (m - 1) * x_stride * 2 + k * 2 >= (1ll << 31))
    return 0;
  if (group_k <= 0 || group_k > k) group_k = k;
  if (!w16 && (!scales || (group_k < k && group_k % 64 != 0))) return 0;  // a piece (64 k of a row) has ONE scale
  SDParams p{};
  p.w = (const char*)w; p.sc = w16 ? nullptr : scales; p.n = (int)n; p.w_stride = (int)w_stride_bytes;
  p.group_n = group_n > 0 ? group_n : 1;
  p.gkb = (int)((group_k + 63) / 64);  // 64-k blocks per scale group (group_k >= k: one group)
  p.s_stride_n = (int)s_stride_n; p.s_stride_k = (int)s_stride_k;
  p.kb_base = pl.kb_base; p.kb_rem = pl.kb_rem; p.RB = pl.RB; p.total = pl.total; p.cap = pl.cap;
  p.m = (int)m; p.x = (const uint16_t*)x; p.out = out; p.x_stride = (int)x_stride;
  hipStream_t st = (hipStream_t)stream;
  const bool two = m > 32;
#define SD_P8(WF, MTT, RR)                                                                        \
  if (pl.P == 2) sd_go<WF, MTT, RR, 2>(p, pl, st); else if (pl.P == 4) sd_go<WF, MTT, RR, 4>(p, pl, st); \
  else if (pl.P == 6) sd_go<WF, MTT, RR, 6>(p, pl, st); else sd_go<WF, MTT, RR, 8>(p, pl, st);
#define SD_P16(WF, MTT, RR)                                                                       \
  if (pl.P == 2) sd_go<WF, MTT, RR, 2>(p, pl, st); else if (pl.P == 4) sd_go<WF, MTT, RR, 4>(p, pl, st); else return 0;
#define SD_R(PM, WF, MTT)                                           \
  switch (pl.R) {                                                   \
    case 1: { PM(WF, MTT, 1) } break;                               \
    case 2: { PM(WF, MTT, 2) } break;                               \
    case 4: { PM(WF, MTT, 4) } break;                               \
    default: return 0;                                              \
  }
#define SD_M(PM, WF) if (two) { SD_R(PM, WF, 2) } else { SD_R(PM, WF, 1) }
  if (wfmt == 1) { SD_M(SD_P8, 1) } else if (wfmt == 2) { SD_M(SD_P8, 2) } else if (wfmt == 4) { SD_M(SD_P16, 4) } else { SD_M(SD_P16, 5) }
#undef SD_M
#undef SD_R
#undef SD_P16
#undef SD_P8
  return hipGetLastError() == hipSuccess ? pl.S : LL_ERR_LAUNCH;
}
