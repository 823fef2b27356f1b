// W4A16 decode GEMM, SHORT-STREAM engine with FINISHED outputs (round 6): narrow projections that must leave fp16 rows -- no
// consumer to add split-K planes up.  Semantics: lite_llama/kernels/quantization/w4a16.py:152-207 (out[m, n] = sum_k x[m, k]
// (nib(n, k) - z) s, fp32 accumulation, + bias, one rounding to fp16); epilogue 1: rows interleaved (gate_j, up_j) ->
// silu(gate) * up on the fp16-rounded sums (kernels/swiglu.py:45-65), as the other W4A16 engines.
//
// Who runs here: the reference-shaped layer (the drop-in route: q, k|v and o projections called one by one through
// w4a16_matmul -- 3584 / 1024 weight rows x 3584: 112 / 32 row groups) and the fused gate|up of a TP >= 4 shard (148 - 296 row
// groups).  The unit loop (gemm_w4_v3.hip) serves them with k-slices + fp32 slabs + a counter merge: 11.7 - 15.3 us for 1.8 - 19 MB,
// 30 - 220 workgroups.  The split-K short-stream engine (gemm_short.hip) cannot: its planes need a consumer.  This body keeps ALL of
// K in one workgroup instead, so the sums meet inside it:
//   * work item = (R row groups of 32 weight rows, one 32-row batch half or the whole batch), all of K; one workgroup per item, one
//     round of workgroups (items <= CUs);
//   * the 8 consumer waves = R row groups x 8 / R k-interleaves request ALL their 1-KB weight pieces (<= 16 per wave) and scale
//     pairs straight into registers at entry, as in gemm_short.hip;
//   * the activation rows cannot stay resident (64 x 3584 fp16 = 448 KB): 4 loader waves stream them through a RING of 128-k
//     chunk tiles in LDS (LDS-DMA, swizzled through the source address), refilled a round behind the consumers -- one s_barrier
//     per round of 8 / R k-blocks; with 32-row batch halves a tile is 8 KB and the ring holds 18 of them (3.5 rounds ahead);
//   * the k-interleaves of a row group meet once through LDS in wave order (deterministic), bias / swiglu in fp32, fp16 stores.
// A launch is bounded by the activation bytes a CU pulls from L2 (224 - 448 KB at ~100 KB / us), not by its weights (57 - 114 KB).
#include <stdlib.h>

#include "common.h"
#include "gemm_w4_common.h"

#define SF_CONSUMERS 8
#define SF_LOADERS 4
#define SF_THREADS ((SF_CONSUMERS + SF_LOADERS) * 64)
#define SF_LDS_BYTES (144 * 1024)  // the ring (and, at the end, the exchange area)

struct SFParams {
  const char* wp;       // packed weights (ll_w4a16_pack_weights)
  const char* sp;       // packed (s, -z s) pairs [K / g][N] x 8 B
  const uint16_t* x;    // [M][K] fp16, row stride x_stride elements
  uint16_t* out;        // [M][N] fp16 (epilogue 0) or [M][N / 2] (epilogue 1)
  const uint16_t* bias; // [N] fp16 or nullptr (epilogue 0)
  int n, chunks, gshift, m, x_stride, epi;
  int halves;           // 1: an item covers the whole batch (MT x 32 rows); 2: items come in pairs of 32-row batch halves (MT = 1)
  int total;            // work items = (N / 32 / R) * halves
};

__device__ __forceinline__ void sf_barrier() { asm volatile("s_barrier" ::: "memory"); }

// at most `n` of this wave's memory operations may still be in flight
__device__ __forceinline__ void sf_wait_outstanding(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
#define SF_C(N) case N: v3_vmcnt<N>(); break;
  switch (n) {
    SF_C(0) SF_C(1) SF_C(2) SF_C(3) SF_C(4) SF_C(5) SF_C(6) SF_C(7) SF_C(8) SF_C(9) SF_C(10) SF_C(11) SF_C(12) SF_C(13) SF_C(14)
    SF_C(15) SF_C(16) SF_C(17) SF_C(18) SF_C(19) SF_C(20) SF_C(21) SF_C(22) SF_C(23) SF_C(24) SF_C(25) SF_C(26) SF_C(27) SF_C(28)
    SF_C(29) SF_C(30) SF_C(31) SF_C(32) SF_C(33) SF_C(34) SF_C(35) SF_C(36) SF_C(37) SF_C(38) SF_C(39) SF_C(40)
    default: if (n < 0) v3_vmcnt<0>(); else v3_vmcnt<40>(); break;
  }
#undef SF_C
}

// MT: 32-row batch tiles per item (1 or 2); R: row groups per item (1 or 2; 8 / R k-interleaves per row group);
// P: pieces (1 KB of packed weights = 32 rows x 64 k) per consumer wave at most (K / 64 / (8 / R) rounded up to 4 / 8 / 12 / 16)
template <int MT, int R, int P>
__global__ __launch_bounds__(SF_THREADS) void wsf_kernel(const SFParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int KQ = SF_CONSUMERS / R;        // k-blocks (64 k) per round = waves per row group
  constexpr int CPR = KQ / 2;                 // 128-k activation chunks per round
  constexpr int XT = MT * 32 * 256;           // one chunk's tile: [MT * 32 rows][16 x 16 B], slot j of row r at j ^ (r & 15)
  constexpr int NSLOT = SF_LDS_BYTES / XT / CPR * CPR;  // ring slots (whole rounds): 18 (MT 1) / 8 (MT 2, R 1) / 9 -> 8 (MT 2, R 2)
  constexpr int PPL = MT * 8 / SF_LOADERS;    // 1-KB pieces (4 rows x 256 B) of a chunk tile per loader wave
  static_assert(NSLOT >= 2 * CPR, "ring: at least two rounds");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = (int)blockIdx.x;
  if (item >= p.total) return;
  const int half = p.halves == 2 ? (item & 1) : 0;
  const int rb = p.halves == 2 ? (item >> 1) : item;
  const int m0 = half * 32;
  const int mrows = p.m - m0 < MT * 32 ? p.m - m0 : MT * 32;  // batch rows of this item (>= 1: the host launches a second half only if m > 32)
  const int nkb = p.chunks * 2;
  const int nC = p.chunks;
  const int rounds = (nkb + KQ - 1) / KQ;

  if (wv >= SF_CONSUMERS) {
    // ============================== activation loaders ============================== //
    const int L = wv - SF_CONSUMERS;
    uint32_t voff[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      int row = (L * PPL + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (row & 15);
      if (row >= mrows) row = mrows - 1;  // rows past the batch feed only unstored outputs
      voff[j] = (uint32_t)((m0 + row) * p.x_stride * 2 + slot * 16);
    }
    const char* xb = (const char*)p.x;
    auto issue = [&](int c) {
#pragma unroll
      for (int j = 0; j < PPL; ++j) v3_dma16<false>((uint32_t)((c % NSLOT) * XT + (L * PPL + j) * 1024), xb + (size_t)c * 256, voff[j]);
    };
    int issued = nC < NSLOT ? nC : NSLOT;
    for (int c = 0; c < issued; ++c) issue(c);
    for (int i = 0; i < rounds; ++i) {
      // round i reads chunks [i CPR, (i + 1) CPR): they have landed when at most the later chunks' operations are in flight
      int need = (i + 1) * CPR;
      if (need > nC) need = nC;
      sf_wait_outstanding((issued - need) * PPL);
      sf_barrier();  // B_i: round i may be read; round i - 1 has been read
      if (i >= 1) {  // refill the slots of round i - 1
        int hi = (i - 1) * CPR + NSLOT + CPR;
        if (hi > nC) hi = nC;
        for (; issued < hi; ++issued) issue(issued);
      }
    }
    return;
  }

  // ================================= consumers ================================= //
  const int r = wv % R, q = wv / R;
  const int rg = rb * R + r;             // row group of 32 weight rows
  const int nl = lane & 31, h = lane >> 5;
  u32x4 w[P];
  u32x2 sc[P];
  {
    const char* wrow = p.wp + ((size_t)(uint32_t)((rg >> 2) * p.chunks) * 8 + (uint32_t)(rg & 3)) * 1024;
    const char* srow = p.sp + (size_t)(uint32_t)rg * 256;
    const uint32_t wl = (uint32_t)lane * 16, sl = (uint32_t)nl * 8;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int kk = i * KQ + q;
      const int kb = kk < nkb ? kk : 0;  // pieces past K re-read block 0 (multiplied by zero; K may have fewer blocks than waves)
      w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)(uint32_t)kb * 4096 + wl));
      sc[i] = *reinterpret_cast<const u32x2*>(srow + (size_t)(uint32_t)((kb >> 1) >> p.gshift) * (uint32_t)p.n * 8 + sl);
    }
  }
  int x_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x_off[j] = nl * 256 + (((h * 4 + j) ^ (nl & 15)) * 16);
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

#pragma unroll
  for (int i = 0; i < P; ++i) {
    if (i < rounds) {  // (wave-uniform, no early exit: the loop unrolls and the weight registers keep static indices; every wave of
                       //  the workgroup passes the same number of barriers)
    sf_barrier();  // B_i
    const int kk = i * KQ + q;
    const bool valid = kk < nkb;
    const int kb = valid ? kk : i * KQ;  // a piece past K: landed data of this round x exact zeros
    const unsigned char* xb = lds + ((kb >> 1) % NSLOT) * XT;
    const int kx = (kb & 1) * 128;
    f16x8 a[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[j][mt] = *reinterpret_cast<const f16x8*>(xb + (x_off[j] ^ kx) + mt * 32 * 256);
    const uint32_t keep = valid ? 0xffffffffu : 0u;
    const uint32_t s0 = sc[i].x & keep, s1 = sc[i].y & keep;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t word = j == 0 ? w[i].x : j == 1 ? w[i].y : j == 2 ? w[i].z : w[i].w;
      const f16x8 wfrag = v3_dequant(word, s0, s1, magic);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, a[j][mt], acc[mt], 0, 0, 0);
    }
    }
  }

  // The KQ partial sums of a row group meet through the LDS the ring no longer needs (the loaders are gone: every chunk has
  // landed before the last round's barrier): [wave][batch row][8 x 16 B], slot j of batch row m at j ^ ((m >> 1) & 7), summed in
  // wave order.
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
    constexpr int QW = MT * 256 / KQ;  // 16-byte quads of the row group's [MT * 32][32] fp32 sums this wave finishes
    const int ncol0 = rg * 32;
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
        if (m < mrows) {
          const int64_t row = m0 + m;
          const int col = ncol0 + quad * 4;
          if (p.epi == 1) {
            // both GEMM outputs rounded to fp16, then silu(g) * u in fp32: the stand-alone kernels' arithmetic
            const float g0 = f16_bits_to_f32(f32_to_f16_bits(v[0])), u0 = f16_bits_to_f32(f32_to_f16_bits(v[1]));
            const float g1 = f16_bits_to_f32(f32_to_f16_bits(v[2])), u1 = f16_bits_to_f32(f32_to_f16_bits(v[3]));
            const uint32_t o2 = (uint32_t)f32_to_f16_bits(ll_silu_mul_f32(g0, u0)) | ((uint32_t)f32_to_f16_bits(ll_silu_mul_f32(g1, u1)) << 16);
            *reinterpret_cast<uint32_t*>(p.out + row * (p.n >> 1) + (col >> 1)) = o2;
          } else {
            if (p.bias) {
              const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + col);
              v[0] += f16_bits_to_f32((uint16_t)(bb.x & 0xffffu));
              v[1] += f16_bits_to_f32((uint16_t)(bb.x >> 16));
              v[2] += f16_bits_to_f32((uint16_t)(bb.y & 0xffffu));
              v[3] += f16_bits_to_f32((uint16_t)(bb.y >> 16));
            }
            const uint32_t lo = (uint32_t)f32_to_f16_bits(v[0]) | ((uint32_t)f32_to_f16_bits(v[1]) << 16);
            const uint32_t hi = (uint32_t)f32_to_f16_bits(v[2]) | ((uint32_t)f32_to_f16_bits(v[3]) << 16);
            *reinterpret_cast<uint2*>(p.out + row * p.n + col) = uint2{lo, hi};
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- //
struct SFPlan {
  int ok = 0, R = 0, MT = 0, halves = 0, P = 0, total = 0;
};

static int sf_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// The plan: one round of workgroups; 32-row batch halves when they fit (half the activation bytes per CU), else the whole batch
// per item; R = 1 when the row groups fit, else 2.  Not served: wide outputs (the row-group engine / unit loop stream those),
// K beyond what a wave's registers hold (16 pieces), an output fed by fewer than 16 row groups (nothing to spread).
static SFPlan sf_plan(int64_t m, int64_t n, int64_t k, int group_size) {
  SFPlan pl;
  static const int on = getenv("LL_GEMM_SF") ? atoi(getenv("LL_GEMM_SF")) : 1;  // A/B knob, read once (0: off)
  if (!on || !ll_w4a16_prepacked_supported(m, n, k, group_size) || m > 64) return pl;
  const int cus = sf_num_cus();
  const int rgs = (int)(n / 32), nkb = (int)(k / 64);
  if (rgs < 16) return pl;
  // candidates in order of preference: batch halves first (half the activation bytes per CU, 8-KB tiles: the ring runs 3 - 8 rounds
  // ahead), then the whole batch with two row groups per item (16-KB tiles, 2-chunk rounds: 3 rounds ahead), last the whole batch
  // with one row group (4-chunk rounds: one round ahead)
  // (a batch of <= 32 rows is one 8-KB-tile item per row block either way: one row group per item first)
  const int cand_big[4][2] = {{1, 2}, {2, 2}, {2, 1}, {1, 1}}, cand_small[4][2] = {{1, 1}, {2, 1}, {0, 0}, {0, 0}};  // (R, halves)
  for (int c = 0; c < 4; ++c) {
    const int R = m > 32 ? cand_big[c][0] : cand_small[c][0], halves = m > 32 ? cand_big[c][1] : cand_small[c][1];
    if (!R || rgs % R) continue;
    const int KQ = 8 / R, pw = (nkb + KQ - 1) / KQ;
    if (pw > 16) continue;
    const int rbs = rgs / R;
    if (rbs * halves > cus) continue;
    pl.ok = 1; pl.R = R; pl.halves = halves; pl.MT = (halves == 2 || m <= 32) ? 1 : 2;
    pl.P = (pw + 3) / 4 * 4; pl.total = rbs * halves;
    return pl;
  }
  return pl;
}

int sf_wants(int64_t m, int64_t n, int64_t k, int group_size, int epilogue) {
  if ((epilogue >> 8) & 3) return 0;  // a forced unit-loop tile width (tests / tuning)
  if ((epilogue & 3) == 2) return 0;  // split-K planes: gemm_short.hip
  return sf_plan(m, n, k, group_size).ok;
}

// Host-side introspection (tests, notebook; no device work): [0] 1 if this engine takes the finished-output launch, [1] work items
// (= grid), [2] R, [3] batch tiles per item, [4] batch halves, [5] pieces per wave (template bound), [6] ring slots, [7] LDS bytes.
extern "C" int ll_w4a16_short_full_plan(int64_t m, int64_t n, int64_t k, int group_size, int32_t* out8) {
  if (!out8) return LL_ERR_ARG;
  const SFPlan pl = sf_plan(m, n, k, group_size);
  const int cpr = pl.ok ? (8 / pl.R) / 2 : 1, xt = pl.ok ? pl.MT * 32 * 256 : 1;
  const int v[8] = {pl.ok, pl.total, pl.R, pl.MT, pl.halves, pl.P, pl.ok ? SF_LDS_BYTES / xt / cpr * cpr : 0, pl.ok ? SF_LDS_BYTES : 0};
  for (int i = 0; i < 8; ++i) out8[i] = v[i];
  return LL_OK;
}

template <int MT, int R, int P>
static void sf_go(const SFParams& p, hipStream_t st) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)wsf_kernel<MT, R, P>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_BYTES);
    attr_set[dev] = true;
  }
  wsf_kernel<MT, R, P><<<dim3((unsigned)p.total), SF_THREADS, SF_LDS_BYTES, st>>>(p);
}

// Shapes / pointers were validated by the caller (v3_launch).
int sf_launch(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias, int64_t m, int64_t n, int64_t k,
              int group_size, int64_t x_stride_m, int epilogue, void* stream) {
  const SFPlan pl = sf_plan(m, n, k, group_size);
  if (!pl.ok) return LL_ERR_SHAPE;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) return LL_ERR_ARG;
  SFParams p{};
  p.wp = (const char*)wpacked; p.sp = (const char*)spacked; p.x = (const uint16_t*)x; p.out = (uint16_t*)out;
  p.bias = (epilogue & 3) == 0 ? (const uint16_t*)bias : nullptr;
  p.n = (int)n; p.chunks = (int)(k / 128); p.m = (int)m; p.x_stride = (int)x_stride_m; p.epi = epilogue & 3;
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  p.halves = pl.halves; p.total = pl.total;
  hipStream_t st = (hipStream_t)stream;
#define SF_P(MTT, RR)                                                                                                   \
  switch (pl.P) {                                                                                                       \
    case 4: sf_go<MTT, RR, 4>(p, st); break;                                                                            \
    case 8: sf_go<MTT, RR, 8>(p, st); break;                                                                            \
    case 12: sf_go<MTT, RR, 12>(p, st); break;                                                                          \
    default: sf_go<MTT, RR, 16>(p, st); break;                                                                          \
  }
  if (pl.MT == 1 && pl.R == 1) SF_P(1, 1)
  else if (pl.MT == 1) SF_P(1, 2)
  else if (pl.R == 1) SF_P(2, 1)
  else SF_P(2, 2)
#undef SF_P
  return LL_LAUNCH_CHECK();
}
