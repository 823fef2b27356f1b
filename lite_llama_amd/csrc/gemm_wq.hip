// Weight-streaming dequant-GEMMs for decode (small M, huge N*K):
//   w4a16_matmul       (a8)  -- reference lite_llama/kernels/quantization/w4a16.py:28-207
//   w8a16_matmul       (a9)  -- reference lite_llama/kernels/quantization/w8a16.py:48-216
//   smoothquant_matmul (a10) -- reference lite_llama/kernels/quantization/w8a8.py:34-217
//
// MI355X design.  The op is HBM-bound on the weight stream (every weight byte is read
// exactly once) with the MFMA + dequant work at M=64 within ~2x of the HBM time, so:
//   * the quantised weights are the MFMA "A" operand, loaded straight from HBM into
//     fragment layout: lane (n = lane&31, h = lane>>5) owns 64 consecutive k of row n
//     (32 B for int4, 64 B for 8-bit) per 128-k unit -- no LDS round trip for the
//     stream, PF units in flight per wave (static register ring, counted vmcnt);
//   * one int32 of the reference int4 format IS one lane's 8-k MFMA fragment: the
//     0x6400 exponent trick turns nibble pairs (j, j+4) into fp16 pairs with one
//     v_and_or each; the activation tile is stored in LDS in the matching k-order;
//   * a dedicated PRODUCER wave stages the activation tile x[64, 128] through a 3-deep
//     LDS ring (vmcnt is per wave: its L2 loads never make the four consumer waves
//     drain their weight ring); consumers read it with conflict-free ds_read_b128;
//   * fp32 accumulation in MFMA 32x32x16 f16 (or exact int32 in 32x32x32 i8 for W8A8);
//   * STREAM-K: the (tile, k-unit) units of the whole GEMM are dealt in equal contiguous
//     ranges to ~2 persistent workgroups per CU, so any N/K shape fills 256 CUs with no
//     tail wave; a tile split across workgroups is combined by its last-arriving wave
//     (per-wave slabs + agent-scope release/acquire, no workgroup barrier, no extra launch).
// hipcc rules learnt the hard way (see DESIGN_NOTEBOOK.md): no control flow around loads in the
// steady-state loop and no use of a loaded value before it is needed -- either makes the
// waitcnt pass drain vmcnt(0) every unit.
#include <stdlib.h>

#include "common.h"

#define GEMM_BN 128        // output columns (weight rows) per workgroup tile: 4 waves x 32
#define GEMM_BM 64         // activation rows per tile (2 MFMA tiles)
#define GEMM_CK 128        // k per unit
#define GEMM_MAX_SLOTS 18  // max workgroups contributing to one tile
#define GEMM_THREADS 320   // waves 0-3 stream weights + MFMA, wave 4 stages the activation tile
#define GEMM_SLAB (GEMM_BN * GEMM_BM)  // fp32 elements of one tile partial

enum { FMT_W4 = 0, FMT_FP8 = 1, FMT_I8 = 2, FMT_I8I8 = 3 };

// 16-byte register quad.  An ext-vector type on purpose: arrays of a plain struct (even alignas(16))
// were left in scratch memory by hipcc, arrays of vector types are promoted to registers.
typedef uint32_t Q4 __attribute__((ext_vector_type(4)));

struct GemmParams {
  void* out;             // fp16 [M, N]
  const void* x;         // fp16 [M, K] (or int8 for W8A8)
  const void* w;         // packed weights [N, ...]
  const float* scales;   // W4: [N, K/g]; W8A16: [ceil(N/gn), ceil(K/gk)]; W8A8: [N]
  const float* zeros;    // W4 only
  const void* bias;      // fp16 [N] or null
  const float* a_scale;  // W8A8: [M]
  int32_t* acc_out;      // W8A8: optional raw int32 accumulators [M, N]
  float* workspace;      // [tiles][slots][GEMM_SLAB] partials
  int32_t* counters;     // [tiles][4] arrived-unit counters (zero on entry, left zero)
  int64_t m, n, k;
  int64_t x_stride, w_stride;  // elements of x per row / int32 words (W4) or bytes of w per row
  int64_t s_stride_n, s_stride_k;
  int group_n;           // W8A16
  int64_t group_k;       // W4: group size; W8A16: min(group_k, K)
  int nblocks, chunks;   // tiles per m-block, 128-k units per tile
  int total_units, upw;  // units overall, units per workgroup
  int slots;             // slab slots per tile
  unsigned gk_magic;     // ceil(2^32 / group_k): k / group_k == umulhi(k, gk_magic)
  unsigned long long* dbg_t;  // optional per-workgroup phase timestamps (LL_GEMM_TRACE), else null
};

// ---------------------------------------------------------------------------------- //
// dequant helpers (fp16 pairs packed in one dword)
// ---------------------------------------------------------------------------------- //
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t pk_fma(uint32_t a, uint32_t b, uint32_t c) {
  f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b),
                                      __builtin_bit_cast(f16x2, c));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_bcast(float v) {
  const uint32_t h = f32_to_f16_bits(v);
  return h | (h << 16);
}
// (w & mask) | magic in ONE VALU op.  hipcc splits it into v_and + v_or when the constants
// are literals (VOP3 takes no literal on gfx9), so the operands are passed in registers.
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask), "v"(magic));
  return r;
}

struct W4Consts {
  uint32_t mask_lo, mask_hi;  // SGPRs
  uint32_t magic;             // VGPR: fp16 1024 | 1024
};

// int4 word -> 4 dwords of fp16 pairs (n0,n4) (n1,n5) (n2,n6) (n3,n7), each = nib * s - z*s.
// Nibble extraction is bit-exact ((w >> 4j) & 0xF, w4a16.py:99-105): the 0x6400 exponent trick
// yields 1024 + nib (or 1024 + 16 nib) exactly, the offset is removed exactly, and the affine
// map is one fp16 fma per pair:  13 VALU per 8 weights.
__device__ __forceinline__ Q4 dequant_w4(uint32_t w, uint32_t s, uint32_t nzs, const W4Consts& k) {
  const uint32_t m1024 = 0xE400E400u;  // -1024.0
  const uint32_t m64 = 0xD400D400u;    // -64.0
  const uint32_t r16 = 0x2C002C00u;    // 1/16
  const uint32_t w2 = w >> 8;
  uint32_t a = and_or(w, k.mask_lo, k.magic);   // 1024 + n0 | 1024 + n4
  uint32_t b = and_or(w, k.mask_hi, k.magic);   // 1024 + 16 n1 | 1024 + 16 n5
  uint32_t c = and_or(w2, k.mask_lo, k.magic);  // n2, n6
  uint32_t d = and_or(w2, k.mask_hi, k.magic);  // n3, n7
  a = pk_add(a, m1024);                         // exact nibble values
  b = pk_fma(b, r16, m64);
  c = pk_add(c, m1024);
  d = pk_fma(d, r16, m64);
  Q4 o;
  o.x = pk_fma(a, s, nzs);
  o.y = pk_fma(b, s, nzs);
  o.z = pk_fma(c, s, nzs);
  o.w = pk_fma(d, s, nzs);
  return o;
}

// 4 int8 -> 2 dwords of fp16 pairs (b0,b1) (b2,b3), exact integer values times s
__device__ __forceinline__ void dequant_i8(uint32_t w, uint32_t s, uint32_t& o0, uint32_t& o1) {
  const uint32_t u = w ^ 0x80808080u;           // b + 128 in [0, 255]
  const uint32_t magic = 0x64646464u;
  uint32_t p0 = __builtin_amdgcn_perm(magic, u, 0x04010400u);  // 1024 + u0 | 1024 + u1
  uint32_t p1 = __builtin_amdgcn_perm(magic, u, 0x04030402u);
  const uint32_t m1152 = 0xE480E480u;           // -1152.0
  o0 = pk_mul(pk_add(p0, m1152), s);
  o1 = pk_mul(pk_add(p1, m1152), s);
}

// 4 fp8-e4m3 bytes -> fp16 pairs via the reference's bit surgery (w8a16.py:48-62):
// bits = ((b & 0x80) << 8) | ((b & 0x7F) << 7)  (value / 256), then * (256 * scale).
__device__ __forceinline__ void dequant_fp8(uint32_t w, uint32_t s256, uint32_t& o0, uint32_t& o1) {
  uint32_t p0 = __builtin_amdgcn_perm(0u, w, 0x010C000Cu);  // b0 << 8 | b1 << 24
  uint32_t p1 = __builtin_amdgcn_perm(0u, w, 0x030C020Cu);
  p0 = (p0 & 0x80008000u) | ((p0 >> 1) & 0x3F803F80u);
  p1 = (p1 & 0x80008000u) | ((p1 >> 1) & 0x3F803F80u);
  o0 = pk_mul(p0, s256);
  o1 = pk_mul(p1, s256);
}

// ---------------------------------------------------------------------------------- //
// The kernel.  grid = ceil(total_units / upw) persistent workgroups, block = 320.
//   FMT : weight format;  MT : 32-row activation tiles (1 or 2);  NSG : scale groups
//   per 64-k lane half (W4 only: 1, 2, 4, 8);  PF : weight units in flight per wave.
// Unit u -> tile = u / chunks (tile = mblk * nblocks + nb), chunk = u % chunks.
// ---------------------------------------------------------------------------------- //
template <int FMT, int MT, int NSG, int PF>
__global__ __launch_bounds__(GEMM_THREADS, 4) void wgemm_kernel(const GemmParams p) {
  constexpr bool INT8A = (FMT == FMT_I8I8);
  constexpr int NQ = (FMT == FMT_W4) ? 2 : 4;                   // Q4 loads per lane per unit
  constexpr int A_ROW_BYTES = INT8A ? (128 + 16) : (256 + 16);  // padded LDS row
  constexpr int A_TILE_BYTES = GEMM_BM * A_ROW_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * A_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int nl = lane & 31, h = lane >> 5;
  const bool producer = (wv == 4);  // wave-uniform role
  const int K = (int)p.k;
  const int chunks = p.chunks;

  const int ub = blockIdx.x * p.upw;
  int ue = ub + p.upw;
  if (ue > p.total_units) ue = p.total_units;
  if (ub >= ue) return;

  // ================================ producer wave ================================== //
  if (producer) {
    // x unit tile = 64 rows x 128 k.  fp16: 16 Q4 per lane, instruction j covers rows
    // 4j..4j+3 with 16 lanes x 16 B = one full 256-B row segment; int8: 8 Q4 per lane.
    constexpr int XQ = INT8A ? 8 : 16;
    constexpr int XLPR = INT8A ? 8 : 16;  // lanes per row
    constexpr int XRPI = 64 / XLPR;       // rows per instruction
    constexpr int KPX = INT8A ? 16 : 8;   // k per Q4
    Q4 xa[XQ];  // one register set: a tile is loaded one full unit before it is stored
    const int xcol = lane % XLPR, xrsub = lane / XLPR;
    int pu = ub, ptile = ub / chunks, pchunk = ub - ptile * chunks;
    int64_t pm0 = (int64_t)(ptile / p.nblocks) * GEMM_BM;
    // Per-lane byte offsets of the 16 (8) rows this lane touches, recomputed only when the
    // m-block changes (never in decode).  Rows >= M are clamped and NOT masked: activation
    // row m only feeds output row m, which the epilogue never stores.  The k tail is not
    // masked either: the consumers zero the WEIGHTS there and the clamped loads read finite x.
    uint32_t roff[XQ];
    auto set_rows = [&]() {
#pragma unroll
      for (int j = 0; j < XQ; ++j) {
        const int64_t row = pm0 + j * XRPI + xrsub;
        const int64_t rc = row < p.m ? row : p.m - 1;
        roff[j] = (uint32_t)(rc * p.x_stride * (INT8A ? 1 : 2));  // x < 4 GiB (checked on the host)
      }
    };
    set_rows();
    auto advance = [&]() {  // next unit, clamped at the last one (re-staging it is harmless)
      if (pu + 1 < ue) {
        ++pu;
        if (++pchunk == chunks) {
          pchunk = 0;
          ++ptile;
          const int64_t nm0 = (int64_t)(ptile / p.nblocks) * GEMM_BM;
          if (nm0 != pm0) {
            pm0 = nm0;
            set_rows();
          }
        }
      }
    };
    const unsigned char* xbase = (const unsigned char*)p.x;
    auto load_x = [&]() {
      const int kk = pchunk * GEMM_CK + xcol * KPX;
      const uint32_t kc = (uint32_t)(kk < K ? kk : K - KPX) * (INT8A ? 1 : 2);
#pragma unroll
      for (int j = 0; j < XQ; ++j) xa[j] = *reinterpret_cast<const Q4*>(xbase + (roff[j] + kc));
    };
    const uint32_t lds_lane = (uint32_t)(xrsub * A_ROW_BYTES + xcol * 16);
    auto store_x = [&](int buf) {
      unsigned char* dst = lds + buf * A_TILE_BYTES + lds_lane;
#pragma unroll
      for (int j = 0; j < XQ; ++j) {
        Q4 v = xa[j];
        if constexpr (FMT == FMT_W4) {
          // [h0..h7] -> (h0,h4) (h1,h5) (h2,h6) (h3,h7): the nibble pairing of dequant_w4
          Q4 t;
          t.x = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);
          t.y = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);
          t.z = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);
          t.w = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);
          v = t;
        }
        *reinterpret_cast<Q4*>(dst + j * XRPI * A_ROW_BYTES) = v;
      }
    };
    // ring prologue: units ub, ub+1 -> LDS buffers 0, 1; unit ub+2 -> registers
    load_x();
    advance();
    store_x(0);
    load_x();
    advance();
    store_x(1);
    load_x();
    advance();
    __syncthreads();
    int wbuf = 2;  // LDS buffer receiving unit u + 2
    for (int u = ub; u < ue; ++u) {
      store_x(wbuf);  // unit u + 2: loaded right after the previous store, a whole unit ago
      load_x();         // unit u + 3
      advance();
      wbuf = wbuf == 2 ? 0 : wbuf + 1;
      __syncthreads();
    }
    return;
  }

  // ================================ consumer waves ================================= //
  unsigned long long* tr = (p.dbg_t && tid == 0) ? p.dbg_t + (size_t)blockIdx.x * 8 : nullptr;
  if (tr) { tr[0] = __builtin_amdgcn_s_memtime(); tr[5] = __builtin_amdgcn_s_memrealtime(); }
  // ---- unit bookkeeping: (tile, chunk) being computed, (ltile, lchunk) being loaded ----
  int tile = ub / chunks, chunk = ub - tile * chunks;
  int lu = ub, ltile = tile, lchunk = chunk;

  // weight row pointer of the tile being LOADED
  const unsigned char* wrow;
  int64_t lrow, srow_off = 0;
  auto set_load_tile = [&](int t) {
    const int nb = t % p.nblocks;
    lrow = (int64_t)nb * GEMM_BN + wv * 32 + nl;
    if (lrow >= p.n) lrow = p.n - 1;  // clamp: loads stay in bounds, stores are masked
    wrow = (const unsigned char*)p.w + lrow * p.w_stride * (FMT == FMT_W4 ? 4 : 1);
    if constexpr (FMT == FMT_FP8 || FMT == FMT_I8) srow_off = (lrow / p.group_n) * p.s_stride_n;
  };
  set_load_tile(ltile);

  Q4 wreg[PF][NQ];
  float sreg[PF][NSG];
  float zreg[PF][NSG];
  auto group_of = [&](int kk) -> int { return (int)__umulhi((unsigned)kk, p.gk_magic); };
  // NOTE: nothing here may *use* a loaded value (not even a select): any use makes hipcc
  // wait for the load right there.  Masking of the k tail happens at consumption time.
  auto load_w = [&](int slot) {
    const int kbase = lchunk * GEMM_CK + h * 64;
    constexpr int KPQ = (FMT == FMT_W4) ? 32 : 16;  // k covered by one Q4
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int kk = kbase + i * KPQ;
      const int kc = kk < K ? kk : K - KPQ;  // K % KPQ == 0 (checked on the host)
      const unsigned char* src = wrow + (FMT == FMT_W4 ? kc / 2 : kc);
      // plain (temporal) load: with one lane per row every 128-B line is touched by 4 loads of
      // 2 consecutive units; a non-temporal hint halves the streaming rate here (measured
      // 2.1 vs 3.7 TB/s, benchmarks/probes/stream_probe.hip)
      wreg[slot][i] = *reinterpret_cast<const Q4*>(src);
    }
    if constexpr (FMT == FMT_W4) {
#pragma unroll
      for (int g = 0; g < NSG; ++g) {
        const int kk = kbase + g * (64 / NSG);
        const int gi = group_of(kk < K ? kk : K - 1);
        sreg[slot][g] = p.scales[lrow * p.s_stride_n + gi];
        zreg[slot][g] = p.zeros[lrow * p.s_stride_n + gi];
      }
    } else if constexpr (FMT == FMT_FP8 || FMT == FMT_I8) {
      const int gi = group_of(kbase < K ? kbase : K - 1);
      sreg[slot][0] = p.scales[srow_off + (int64_t)gi * p.s_stride_k];
    }
  };
  auto advance_load = [&]() {  // clamped at the last unit: the redundant re-loads hit L2/TCP
    if (lu + 1 < ue) {
      ++lu;
      if (++lchunk == chunks) {
        lchunk = 0;
        ++ltile;
        set_load_tile(ltile);  // address math only, no loads inside this branch
      }
    }
  };

  // ---- accumulators ----
  f32x16 accf[MT];
  i32x16 acci[MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accf[mt][r] = 0.f;
        acci[mt][r] = 0;
      }
    }
  };
  zero_acc();

  W4Consts kc4;
  kc4.mask_lo = 0x000F000Fu;
  kc4.mask_hi = 0x00F000F0u;
  kc4.magic = 0x64006400u;
  asm volatile("" : "+v"(kc4.magic));  // keep it in a VGPR (VOP3 constant-bus limit)

  auto compute = [&](int slot, int buf, int c) {
    const int kbase = c * GEMM_CK + h * 64;
    const unsigned char* abase = lds + buf * A_TILE_BYTES + nl * A_ROW_BYTES + h * (INT8A ? 64 : 128);
    if constexpr (INT8A) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Q4 wq = wreg[slot][s];
        const uint32_t wm = (kbase + s * 16) < K ? 0xFFFFFFFFu : 0u;  // k tail must be exactly zero
        wq.x &= wm; wq.y &= wm; wq.z &= wm; wq.w &= wm;
        const i32x4 wfrag = __builtin_bit_cast(i32x4, wq);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const i32x4 afrag = *reinterpret_cast<const i32x4*>(abase + mt * 32 * A_ROW_BYTES + s * 16);
          acci[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfrag, afrag, acci[mt], 0, 0, 0);
        }
      }
    } else {
      uint32_t sp[NSG], zp[NSG];
#pragma unroll
      for (int g = 0; g < NSG; ++g) {
        // zero scale -> zero weight for the k tail (the clamped loads read valid, finite data).
        // Integer AND masks, not selects: hipcc turns a float select into an exec-masked branch.
        const uint32_t gm = (kbase + g * (64 / NSG)) < K ? 0xFFFFFFFFu : 0u;
        if constexpr (FMT == FMT_W4) {
          sp[g] = pk_bcast(sreg[slot][g]) & gm;
          zp[g] = pk_bcast(-zreg[slot][g] * sreg[slot][g]) & gm;  // -(z*s): fp32 product, one rounding
        } else if constexpr (FMT == FMT_FP8) {
          sp[g] = pk_bcast(sreg[slot][g] * 256.0f) & gm;
          zp[g] = 0;
        } else {
          sp[g] = pk_bcast(sreg[slot][g]) & gm;
          zp[g] = 0;
        }
      }
      // A fragments are read one k-step ahead of their MFMAs (LDS latency ~100+ cycles would
      // otherwise sit in front of every MFMA pair).
      f16x8 acur[MT], anxt[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acur[mt] = *reinterpret_cast<const f16x8*>(abase + mt * 32 * A_ROW_BYTES);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            anxt[mt] = *reinterpret_cast<const f16x8*>(abase + mt * 32 * A_ROW_BYTES + (s + 1) * 16);
        }
        Q4 wf;
        if constexpr (FMT == FMT_W4) {
          const Q4& src = wreg[slot][s >> 2];
          const uint32_t word = (s & 3) == 0 ? src.x : (s & 3) == 1 ? src.y : (s & 3) == 2 ? src.z : src.w;
          constexpr int WPG = 8 / NSG;  // words per scale group
          wf = dequant_w4(word, sp[s / WPG], zp[s / WPG], kc4);
        } else {
          const Q4& src = wreg[slot][s >> 1];
          const uint32_t w0 = (s & 1) ? src.z : src.x;
          const uint32_t w1 = (s & 1) ? src.w : src.y;
          uint32_t d0, d1, d2, d3;
          if constexpr (FMT == FMT_FP8) {
            dequant_fp8(w0, sp[0], d0, d1);
            dequant_fp8(w1, sp[0], d2, d3);
          } else {
            dequant_i8(w0, sp[0], d0, d1);
            dequant_i8(w1, sp[0], d2, d3);
          }
          wf = Q4{d0, d1, d2, d3};
        }
        const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          accf[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, acur[mt], accf[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead reads ahead of this step's MFMAs
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acur[mt] = anxt[mt];
      }
    }
  };

  // ---- tile flush: direct epilogue, or publish a partial and let the last arriver finish ----
  // Entirely wave-local (each wave owns 32 output columns): no workgroup barrier.
  auto flush = [&](int t, int c_lo, int c_hi /* inclusive */) {
    const int mblk = t / p.nblocks, nb = t - mblk * p.nblocks;
    const int64_t m0 = (int64_t)mblk * GEMM_BM;
    const bool full = (c_lo == 0 && c_hi == chunks - 1);
    if (!full) {
      // contributors of tile t are the consecutive workgroups covering its unit range
      const int w0 = (int)(((int64_t)t * chunks) / p.upw);
      const int slot = (int)blockIdx.x - w0;
      float* ws = p.workspace + (((int64_t)t * p.slots + slot) * 4 + wv) * (GEMM_SLAB / 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          if constexpr (INT8A)
            v = __builtin_bit_cast(f32x4, i32x4{acci[mt][4 * g], acci[mt][4 * g + 1], acci[mt][4 * g + 2],
                                                acci[mt][4 * g + 3]});
          else
            v = f32x4{accf[mt][4 * g], accf[mt][4 * g + 1], accf[mt][4 * g + 2], accf[mt][4 * g + 3]};
          // write-through (sc1) store: nothing stays dirty in this XCD's L2, so no L2
          // write-back (release fence) is needed before the ticket -- that fence costs 5-25 us
          // when hundreds of workgroups flush at once (cdna guide G16 recipe R1).
          float* dst = ws + ((mt * 4 + g) * 64 + lane) * 4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
      }
      // publish: every lane drains its stores, then ONE relaxed agent-scope ticket
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int old = 0;
      if (lane == 0) {
        old = __hip_atomic_fetch_add(&p.counters[t * 4 + wv], c_hi - c_lo + 1, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
      }
      old = __builtin_amdgcn_readfirstlane(old);
      if (old + (c_hi - c_lo + 1) != chunks) return;  // someone else finishes this tile
      if (lane == 0)
        __hip_atomic_store(&p.counters[t * 4 + wv], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      zero_acc();
      const int w1 = (int)((((int64_t)t + 1) * chunks - 1) / p.upw);
      for (int sl = 0; sl <= w1 - w0; ++sl) {
        const float* wr = p.workspace + (((int64_t)t * p.slots + sl) * 4 + wv) * (GEMM_SLAB / 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + ((mt * 4 + g) * 64 + lane) * 4);
            if constexpr (INT8A) {
              const i32x4 iv = __builtin_bit_cast(i32x4, v);
#pragma unroll
              for (int e = 0; e < 4; ++e) acci[mt][4 * g + e] += iv[e];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) accf[mt][4 * g + e] += v[e];
            }
          }
        }
      }
    }
    // ---- epilogue: D[n][m], lane = column m (nl), rows n = 8g + 4h + e ----
    const uint16_t* bias = (const uint16_t*)p.bias;
    uint16_t* out = (uint16_t*)p.out;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t mrow = m0 + nl + mt * 32;
      if (mrow >= p.m) continue;
      float asc = 1.f;
      if constexpr (INT8A) asc = p.a_scale[mrow];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t nn = (int64_t)nb * GEMM_BN + wv * 32 + 8 * g + 4 * h;
        uint16_t o[4];
        float bv[4], wsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t nc = (nn + e) < p.n ? (nn + e) : (p.n - 1);  // clamped, branch-free loads
          bv[e] = bias ? f16_bits_to_f32(bias[nc]) : 0.f;
          wsv[e] = INT8A ? p.scales[nc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t ncol = nn + e;
          float v;
          if constexpr (INT8A) {
            const int32_t a = acci[mt][4 * g + e];
            if (p.acc_out && ncol < p.n) p.acc_out[mrow * p.n + ncol] = a;
            // (acc.f32 * a_scale[m]) * w_scale[n]  (w8a8.py:118-120)
            v = ((float)a * asc) * wsv[e];
          } else {
            v = accf[mt][4 * g + e];
          }
          v += bv[e];
          o[e] = f32_to_f16_bits(v);
        }
        if (nn + 3 < p.n && (p.n & 3) == 0) {
          uint2 pk;
          pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
          *reinterpret_cast<uint2*>(out + mrow * p.n + nn) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nn + e < p.n) out[mrow * p.n + nn + e] = o[e];
        }
      }
    }
  };

  // ---- main loop ----
  // PF statically unrolled bodies (static ring slots) with early-exit breaks; every load is
  // unconditional + clamped, so the waitcnt pass keeps exact counted vmcnt in steady state.
  // A tile/range end breaks out to ONE copy of the flush code, after which the ring is
  // rotated (register moves, once per tile) so the next unit sits in slot 0 again.
#pragma unroll
  for (int pf = 0; pf < PF; ++pf) {
    load_w(pf);
    advance_load();
  }
  __syncthreads();
  if (tr) tr[1] = __builtin_amdgcn_s_memtime();
  int u = ub, rbuf = 0, seg_lo = chunk;
  unsigned long long cwt = 0;
  for (;;) {
    int rot = PF;
    bool seg_end = false;
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
      compute(pf, rbuf, chunk);
      load_w(pf);
      advance_load();
      rbuf = rbuf == 2 ? 0 : rbuf + 1;
      const unsigned long long b0 = tr ? __builtin_amdgcn_s_memtime() : 0;
      __syncthreads();
      if (tr) cwt += __builtin_amdgcn_s_memtime() - b0;
      ++u;
      if (chunk == chunks - 1 || u >= ue) {
        seg_end = true;
        rot = pf + 1;
        break;
      }
      ++chunk;
    }
    if (seg_end) {
      if (tr) tr[2] = __builtin_amdgcn_s_memtime();
      flush(tile, seg_lo, chunk);
      if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[3] = __builtin_amdgcn_s_memtime(); tr[4] = (unsigned long long)(ue - ub); tr[7] = cwt; tr[6] = __builtin_amdgcn_s_memrealtime(); }
      if (u >= ue) break;
      zero_acc();
      ++tile;
      chunk = 0;
      seg_lo = 0;
      // rotate the ring left by `rot` so that slot 0 holds unit u again
      for (int r = 0; r < (rot % PF); ++r) {
        Q4 t0[NQ];
        float ts[NSG], tz[NSG];
#pragma unroll
        for (int i = 0; i < NQ; ++i) t0[i] = wreg[0][i];
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
          ts[g] = sreg[0][g];
          tz[g] = zreg[0][g];
        }
#pragma unroll
        for (int q = 0; q + 1 < PF; ++q) {
#pragma unroll
          for (int i = 0; i < NQ; ++i) wreg[q][i] = wreg[q + 1][i];
#pragma unroll
          for (int g = 0; g < NSG; ++g) {
            sreg[q][g] = sreg[q + 1][g];
            zreg[q][g] = zreg[q + 1][g];
          }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) wreg[PF - 1][i] = t0[i];
#pragma unroll
        for (int g = 0; g < NSG; ++g) {
          sreg[PF - 1][g] = ts[g];
          zreg[PF - 1][g] = tz[g];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct Plan {
  int mblocks, nblocks, chunks, total_units, upw, grid, slots;
};

static int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static Plan make_plan(int64_t m, int64_t n, int64_t k) {
  Plan pl;
  pl.mblocks = (int)((m + GEMM_BM - 1) / GEMM_BM);
  pl.nblocks = (int)((n + GEMM_BN - 1) / GEMM_BN);
  pl.chunks = (int)((k + GEMM_CK - 1) / GEMM_CK);
  pl.total_units = pl.mblocks * pl.nblocks * pl.chunks;
  int target = 2 * num_cus();  // ~2 persistent workgroups per CU
  if (const char* e = getenv("LL_GEMM_WGS")) {  // tuning knob
    const int v = atoi(e);
    if (v > 0) target = v;
  }
  int upw = (pl.total_units + target - 1) / target;
  // a tile may be shared by at most GEMM_MAX_SLOTS workgroups (workspace bound)
  const int min_upw = (pl.chunks + (GEMM_MAX_SLOTS - 2) - 1) / (GEMM_MAX_SLOTS - 2);
  if (upw < min_upw) upw = min_upw;
  if (upw < 2) upw = 2;
  pl.upw = upw;
  pl.grid = (pl.total_units + upw - 1) / upw;
  int slots = (pl.chunks - 1) / upw + 2;
  if (slots > GEMM_MAX_SLOTS) slots = GEMM_MAX_SLOTS;
  pl.slots = slots;
  return pl;
}

extern "C" int ll_w4a16_v3_workspace(int64_t m, int64_t n, int64_t k, int64_t* floats, int64_t* ints);  // gemm_w4_v3.hip

// the decode-shaped 8-bit engine (gemm_w8_skinny.hip): served shapes leave S fp32 / int32 partial planes in the workspace
extern "C" int64_t ll_dense8_partial_words(int64_t m, int64_t n, int64_t k);
extern "C" int64_t ll_dense16_partial_words(int64_t m, int64_t n, int64_t k);
extern "C" int ll_w8_mtiled_try(void* out, const void* x, const void* w, const float* scales, const float* a_scale, const void* bias,
                                int32_t* acc_out, int64_t m, int64_t n, int64_t k, int group_n, int64_t group_k, int wfmt,
                                int64_t x_stride, int64_t w_stride, int64_t s_stride_n, int64_t s_stride_k, void* stream);  // gemm_w8_prefill.hip
extern "C" int ll_dense8_try(void* out, const void* x, const void* w, const float* scales, const float* a_scale,
                             const void* bias, int32_t* acc_out, int64_t m, int64_t n, int64_t k, int group_n,
                             int64_t group_k, int wfmt, int64_t x_stride, int64_t w_stride, int64_t s_stride_n,
                             int64_t s_stride_k, void* partials, void* stream);

extern "C" int ll_gemm_workspace(int64_t m, int64_t n, int64_t k, int64_t* workspace_floats,
                                 int64_t* counter_ints) {
  const Plan pl = make_plan(m, n, k);
  const int64_t tiles = (int64_t)pl.mblocks * pl.nblocks;
  int64_t f = tiles * pl.slots * GEMM_SLAB, c = tiles * 4, f2 = 0, c2 = 0;
  ll_w4a16_v3_workspace(m, n, k, &f2, &c2);  // the scratch must fit whichever engine is dispatched
  const int64_t f3 = ll_dense8_partial_words(m, n, k);
  if (f3 > f) f = f3;
  const int64_t f4 = ll_dense16_partial_words(m, n, k);
  if (f4 > f) f = f4;
  if (workspace_floats) *workspace_floats = f > f2 ? f : f2;
  if (counter_ints) *counter_ints = c > c2 ? c : c2;
  return LL_OK;
}

template <int FMT, int NSG>
static int launch_wgemm(GemmParams& p, hipStream_t st) {
  const Plan pl = make_plan(p.m, p.n, p.k);
  p.nblocks = pl.nblocks;
  p.chunks = pl.chunks;
  p.total_units = pl.total_units;
  p.upw = pl.upw;
  p.slots = pl.slots;
  p.gk_magic = p.group_k >= ((int64_t)1 << 31)
                   ? 0u
                   : (unsigned)((((uint64_t)1 << 32) + p.group_k - 1) / (uint64_t)p.group_k);
  if (!p.workspace || !p.counters) return LL_ERR_ARG;
  p.dbg_t = nullptr;
  if (const char* e = getenv("LL_GEMM_TRACE")) p.dbg_t = (unsigned long long*)strtoull(e, nullptr, 0);
  dim3 grid((unsigned)pl.grid);
  constexpr int PF = (FMT == FMT_W4) ? 4 : 2;
  if (p.m <= 32)
    wgemm_kernel<FMT, 1, NSG, PF><<<grid, GEMM_THREADS, 0, st>>>(p);
  else
    wgemm_kernel<FMT, 2, NSG, PF><<<grid, GEMM_THREADS, 0, st>>>(p);
  return LL_LAUNCH_CHECK();
}

// Reference-layout entry (any M, any group size): the generic engine.  Decode-shaped calls reach the pre-packed engine
// (gemm_w4_v3.hip) through the Python mirror, which builds the load-time layout once per weight.
extern "C" int ll_w4a16_matmul(void* out, const void* x, const int32_t* qweight, const float* scales,
                               const float* zeros, const void* bias, int64_t m, int64_t n, int64_t k,
                               int group_size, int64_t x_stride_m, int64_t qw_stride_n,
                               int64_t s_stride_n, float* workspace, int32_t* counters, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || group_size <= 0 || k % group_size != 0) return LL_ERR_SHAPE;
  if (k % 32 != 0 || x_stride_m % 8 != 0 || qw_stride_n % 4 != 0) return LL_ERR_SHAPE;
  if (!ll_aligned16(x) || !ll_aligned16(qweight)) return LL_ERR_ARG;
  if (m == 0) return LL_OK;
  GemmParams p{};
  p.out = out; p.x = x; p.w = qweight; p.scales = scales; p.zeros = zeros; p.bias = bias;
  p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m; p.w_stride = qw_stride_n;
  p.s_stride_n = s_stride_n; p.s_stride_k = 1; p.group_n = 1; p.group_k = group_size;
  hipStream_t st = (hipStream_t)stream;
  if (group_size % 64 == 0) return launch_wgemm<FMT_W4, 1>(p, st);
  if (group_size == 32) return launch_wgemm<FMT_W4, 2>(p, st);
  if (group_size == 16) return launch_wgemm<FMT_W4, 4>(p, st);
  if (group_size == 8) return launch_wgemm<FMT_W4, 8>(p, st);
  return LL_ERR_SHAPE;
}

extern "C" int ll_w8a16_matmul(void* out, const void* x, const void* qweight, const float* scales,
                               const void* bias, int64_t m, int64_t n, int64_t k, int group_n,
                               int64_t group_k, int wfmt, int64_t x_stride_m, int64_t qw_stride_n,
                               int64_t s_stride_n, int64_t s_stride_k, float* workspace,
                               int32_t* counters, void* stream) {
  if (wfmt != LL_W_FP8E4M3 && wfmt != LL_W_INT8) return LL_ERR_DTYPE;
  if (m < 0 || n <= 0 || k <= 0 || group_n <= 0 || group_k <= 0) return LL_ERR_SHAPE;
  if (group_k > k) group_k = k;
  if (group_k % 128 != 0 && group_k < k) return LL_ERR_SHAPE;  // w8a16.py:189-191
  if (k % 16 != 0 || x_stride_m % 8 != 0 || qw_stride_n % 16 != 0) return LL_ERR_SHAPE;
  if (!ll_aligned16(x) || !ll_aligned16(qweight)) return LL_ERR_ARG;
  if (m == 0) return LL_OK;
  if (m > 64) {  // prefill shapes (round 6): M x N tiled MFMA GEMM, the weight tile fetched once per 256 rows (gemm_w8_prefill.hip)
    const int r = ll_w8_mtiled_try(out, x, qweight, scales, nullptr, bias, nullptr, m, n, k, group_n, group_k,
                                   wfmt == LL_W_FP8E4M3 ? 1 : 2, x_stride_m, qw_stride_n, s_stride_n, s_stride_k, stream);
    if (r != 0) return r > 0 ? LL_OK : r;
  }
  {  // decode shapes: full-line weight tiles, split-K partials in the workspace (gemm_w8_skinny.hip)
    const int r = ll_dense8_try(out, x, qweight, scales, nullptr, bias, nullptr, m, n, k, group_n, group_k,
                                wfmt == LL_W_FP8E4M3 ? 1 : 2, x_stride_m, qw_stride_n, s_stride_n, s_stride_k, workspace,
                                stream);
    if (r != 0) return r > 0 ? LL_OK : r;
  }
  GemmParams p{};
  p.out = out; p.x = x; p.w = qweight; p.scales = scales; p.bias = bias;
  p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m; p.w_stride = qw_stride_n;
  p.s_stride_n = s_stride_n; p.s_stride_k = s_stride_k; p.group_n = group_n; p.group_k = group_k;
  hipStream_t st = (hipStream_t)stream;
  return wfmt == LL_W_FP8E4M3 ? launch_wgemm<FMT_FP8, 1>(p, st) : launch_wgemm<FMT_I8, 1>(p, st);
}

extern "C" int ll_w8a8_matmul(void* out, const int8_t* qa, const float* a_scale, const int8_t* qweight,
                              const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                              int64_t qw_stride_n, int32_t* acc_out, int32_t* workspace,
                              int32_t* counters, void* stream) {
  if (m < 0 || n <= 0 || k <= 0) return LL_ERR_SHAPE;
  if (k % 16 != 0 || qw_stride_n % 16 != 0) return LL_ERR_SHAPE;
  if (!ll_aligned16(qa) || !ll_aligned16(qweight)) return LL_ERR_ARG;
  if (m == 0) return LL_OK;
  if (m > 64) {  // prefill shapes (round 6): the M-tiled int8 x int8 engine (gemm_w8_prefill.hip)
    const int r = ll_w8_mtiled_try(out, qa, qweight, w_scale, a_scale, bias, acc_out, m, n, k, 1, k, 3, k, qw_stride_n, 0, 0, stream);
    if (r != 0) return r > 0 ? LL_OK : r;
  }
  {
    const int r = ll_dense8_try(out, qa, qweight, w_scale, a_scale, bias, acc_out, m, n, k, 1, k, 3, k, qw_stride_n, 0, 0,
                                workspace, stream);
    if (r != 0) return r > 0 ? LL_OK : r;
  }
  GemmParams p{};
  p.out = out; p.x = qa; p.w = qweight; p.scales = w_scale; p.bias = bias; p.a_scale = a_scale;
  p.acc_out = acc_out; p.workspace = (float*)workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = k; p.w_stride = qw_stride_n;
  p.group_n = 1; p.group_k = k;
  return launch_wgemm<FMT_I8I8, 1>(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- //
// per-token activation quantiser -- reference w8a8.py:34-68.  One block per row.
// scale = absmax / 127 (1.0 if 0); q = trunc(x / scale) (float -> int8 cast truncates).
// ---------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void quant_act_kernel(int8_t* __restrict__ q, float* __restrict__ a_scale,
                                                        const uint16_t* __restrict__ x, int64_t k,
                                                        int64_t x_stride) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const uint16_t* xr = x + row * x_stride;
  float amax = 0.f;
  for (int64_t i = threadIdx.x; i < k; i += 256) amax = fmaxf(amax, fabsf(f16_bits_to_f32(xr[i])));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float scale = amax / 127.0f;
  scale = scale > 0.f ? scale : 1.0f;
  for (int64_t i = threadIdx.x; i < k; i += 256) {
    const float v = f16_bits_to_f32(xr[i]) / scale;
    q[row * k + i] = (int8_t)(int)truncf(v);
  }
  if (threadIdx.x == 0) a_scale[row] = scale;
}

extern "C" int ll_quant_act_cached_try(int8_t* q, float* a_scale, const void* x, int64_t m, int64_t k, int64_t x_stride_m,
                                       void* stream);
extern "C" int ll_quantize_activations_int8(int8_t* q, float* a_scale, const void* x, int64_t m, int64_t k,
                                            int64_t x_stride_m, void* stream) {
  if (m < 0 || k <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  // decode-sized rows: one pass with the row in registers (w8a8_fused.hip); anything else: the two-pass kernel above
  if (const int rc = ll_quant_act_cached_try(q, a_scale, x, m, k, x_stride_m, stream)) return rc < 0 ? rc : LL_OK;
  quant_act_kernel<<<dim3((unsigned)m), 256, 0, (hipStream_t)stream>>>(q, a_scale, (const uint16_t*)x, k,
                                                                        x_stride_m);
  return LL_LAUNCH_CHECK();
}
