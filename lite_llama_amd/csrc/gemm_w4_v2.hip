// W4A16 dequant-GEMM, second-generation engine (group_size % 128 == 0: the AWQ/GPTQ g128 case).
// Reference semantics: lite_llama/kernels/quantization/w4a16.py:28-207 (see gemm_wq.hip for the
// generic engine that also serves the other group sizes and the 8-bit formats).
//
// What round-1 measurements (benchmarks/gemm_trace.py, probes/stream_probe.hip) said
// about the first engine, and what this one does about it:
//   * lane-per-row weight loads stream at 3.7 TB/s, 64-B-per-row coalesced loads at 4.3+ TB/s
//     -> dedicated LOADER waves fetch the [128 rows x 64 B] unit tile with 4 lanes per row and stage
//        it (XOR-swizzled, conflict-free) in an LDS ring; scale/zero pairs ride along;
//   * one consumer wave per SIMD is issue-bound (~4 cycles/instruction) -> 8 consumer waves
//     (2 per SIMD): 4 row groups x 2 k-halves of every unit, each doing 4 MFMA steps;
//     the two k-halves are summed through LDS once per tile;
//   * per-workgroup fixed costs dominated (cold prologue, many-contributor tile merges)
//     -> ONE persistent 12-wave workgroup per CU, one HBM round trip in the prologue,
//        <= 3 contributors per tile.
// Roles (wave id): 0-7 consumers (ng = w & 3, kh = w >> 2), 8-9 weight loaders (64 rows each),
// 10-11 activation producers (32 rows each).  One s_barrier per unit for everybody.
#include <stdlib.h>

#include "common.h"

#define V2_BN 128
#define V2_BM 64
#define V2_CK 128
#define V2_THREADS 768
#define V2_SLAB (V2_BN * V2_BM)
#define V2_MAX_SLOTS 6
#define V2_WPF 6  // weight units in flight (registers) per loader wave
#define V2_RW 4   // LDS weight ring depth (units)
#define V2_D 2    // a unit is staged in LDS this many iterations before it is consumed

struct alignas(16) Q4 {
  uint32_t x, y, z, w;
};

struct V2Params {
  uint16_t* out;
  const uint16_t* x;
  const uint32_t* w;
  const float* scales;
  const float* zeros;
  const uint16_t* bias;
  float* workspace;
  int32_t* counters;
  int64_t m, n, k;
  int64_t x_stride, w_stride, s_stride;  // elements / int32 words / floats per row
  int nblocks, chunks, total_units, upw, slots;
  int gshift;  // log2(group_size / 128) when a power of two, else -1
  int gdiv;    // group_size / 128
  unsigned long long* dbg;  // optional phase stamps (LL_GEMM_TRACE)
  int ablate;               // LL_GEMM2_ABLATE: 1 no x loads, 2 no weight loads, 4 no compute (debug)
};

__device__ __forceinline__ uint32_t v2_pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t v2_pk_fma(uint32_t a, uint32_t b, uint32_t c) {
  f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b),
                                      __builtin_bit_cast(f16x2, c));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t v2_bcast(float v) {
  const uint32_t h = f32_to_f16_bits(v);
  return h | (h << 16);
}
__device__ __forceinline__ uint32_t v2_and_or(uint32_t w, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask), "v"(magic));
  return r;
}
// bit-exact nibble unpack (w4a16.py:99-105) + affine map, 13 VALU per 8 weights (see gemm_wq.hip)
__device__ __forceinline__ Q4 v2_dequant(uint32_t w, uint32_t s, uint32_t nzs, uint32_t magic) {
  const uint32_t w2 = w >> 8;
  uint32_t a = v2_and_or(w, 0x000F000Fu, magic);
  uint32_t b = v2_and_or(w, 0x00F000F0u, magic);
  uint32_t c = v2_and_or(w2, 0x000F000Fu, magic);
  uint32_t d = v2_and_or(w2, 0x00F000F0u, magic);
  a = v2_pk_add(a, 0xE400E400u);
  b = v2_pk_fma(b, 0x2C002C00u, 0xD400D400u);
  c = v2_pk_add(c, 0xE400E400u);
  d = v2_pk_fma(d, 0x2C002C00u, 0xD400D400u);
  Q4 o;
  o.x = v2_pk_fma(a, s, nzs);
  o.y = v2_pk_fma(b, s, nzs);
  o.z = v2_pk_fma(c, s, nzs);
  o.w = v2_pk_fma(d, s, nzs);
  return o;
}

// LDS map (bytes)
#define V2_A_ROW 272
#define V2_A_TILE (V2_BM * V2_A_ROW)              // 17408
#define V2_XR 4                                   // LDS activation ring depth (units)
#define V2_OFF_A 0                                // V2_XR x-tiles
#define V2_OFF_W (V2_XR * V2_A_TILE)              // RW weight tiles of 8192 B
#define V2_OFF_S (V2_OFF_W + V2_RW * 8192)        // RW scale tiles: 128 rows x (s, -z*s) as fp16 pairs x2 = 8 B
#define V2_OFF_R (V2_OFF_S + V2_RW * 1024)        // reduce buffer: 4 row groups x 8 KB
#define V2_LDS_BYTES (V2_OFF_R + 4 * 8192)

// swizzled byte offset of 16-B piece p (0..3) of row r (0..127) inside an 8-KB weight tile
__device__ __forceinline__ int v2_wslot(int r, int p) { return (r * 4 + (p ^ ((r >> 2) & 3))) * 16; }

#define V2_TBAR()                                                  \
  {                                                                \
    const unsigned long long b0_ = dbgp ? __builtin_amdgcn_s_memtime() : 0; \
    __syncthreads();                                               \
    if (dbgp) bwait += __builtin_amdgcn_s_memtime() - b0_;          \
  }

template <int MT>
__global__ __launch_bounds__(V2_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgemm2_kernel(const V2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int K = (int)p.k;
  const int chunks = p.chunks;
  const int ub = blockIdx.x * p.upw;
  int ue = ub + p.upw;
  if (ue > p.total_units) ue = p.total_units;
  if (ub >= ue) return;

  // every role walks the same (tile, chunk) sequence, so the seg_end barriers always match
  const int tile0 = ub / chunks, chunk0 = ub - tile0 * chunks;
  // debug stamps: role slots 0 (consumer wave 0), 1 (loader wave 8), 2 (producer wave 10)
  const int drole = wv == 0 ? 0 : wv == 8 ? 1 : wv == 10 ? 2 : -1;
  unsigned long long* dbgp = (p.dbg && lane == 0 && drole >= 0) ? p.dbg + ((size_t)blockIdx.x * 3 + drole) * 4 : nullptr;
  unsigned long long bwait = 0;
  const unsigned long long tstart = dbgp ? __builtin_amdgcn_s_memtime() : 0;
  auto group_of = [&](int c) -> int { return p.gshift >= 0 ? (c >> p.gshift) : (c / p.gdiv); };

  if (wv >= 10) {
    // =============================== activation producers =============================== //
    const int half = wv - 10;  // rows half*32 .. +31
    const int xcol = lane & 15, xrsub = lane >> 4;
    int pu = ub, ptile = tile0, pchunk = chunk0;
    int64_t pm0 = (int64_t)(ptile / p.nblocks) * V2_BM;
    uint32_t roff[8];
    auto set_rows = [&]() {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t row = pm0 + half * 32 + j * 4 + xrsub;
        const int64_t rc = row < p.m ? row : p.m - 1;  // rows >= M feed only unstored outputs
        roff[j] = (uint32_t)(rc * p.x_stride * 2);
      }
    };
    set_rows();
    auto advance = [&]() {
      if (pu + 1 < ue) {
        ++pu;
        if (++pchunk == chunks) {
          pchunk = 0;
          ++ptile;
          const int64_t nm0 = (int64_t)(ptile / p.nblocks) * V2_BM;
          if (nm0 != pm0) {
            pm0 = nm0;
            set_rows();
          }
        }
      }
    };
    const unsigned char* xbase = (const unsigned char*)p.x;
    const uint32_t lds_lane = (uint32_t)((half * 32 + xrsub) * V2_A_ROW + xcol * 16);
#define V2_DECL_X(P) i32x4 P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7
#define V2_LOAD_X1(P, J) P##J = *reinterpret_cast<const i32x4*>(xbase + (roff[J] + kc_));
#define V2_LOAD_X(P)                                                                        \
  if (!(p.ablate & 1)) {                                                                    \
    const int kk_ = pchunk * V2_CK + xcol * 8;                                              \
    const uint32_t kc_ = (uint32_t)(kk_ < K ? kk_ : K - 8) * 2; /* k tail: weights zeroed */ \
    V2_LOAD_X1(P, 0) V2_LOAD_X1(P, 1) V2_LOAD_X1(P, 2) V2_LOAD_X1(P, 3)                     \
    V2_LOAD_X1(P, 4) V2_LOAD_X1(P, 5) V2_LOAD_X1(P, 6) V2_LOAD_X1(P, 7)                     \
    advance();                                                                              \
  } else { advance(); }
// [h0..h7] -> (h0,h4) (h1,h5) (h2,h6) (h3,h7): the nibble pairing of the dequant
#define V2_STORE_X1(P, J, DST)                                                       \
  {                                                                                  \
    const i32x4 v_ = P##J;                                                           \
    i32x4 t_;                                                                        \
    t_.x = (int)__builtin_amdgcn_perm((uint32_t)v_.z, (uint32_t)v_.x, 0x05040100u);  \
    t_.y = (int)__builtin_amdgcn_perm((uint32_t)v_.z, (uint32_t)v_.x, 0x07060302u);  \
    t_.z = (int)__builtin_amdgcn_perm((uint32_t)v_.w, (uint32_t)v_.y, 0x05040100u);  \
    t_.w = (int)__builtin_amdgcn_perm((uint32_t)v_.w, (uint32_t)v_.y, 0x07060302u);  \
    *reinterpret_cast<i32x4*>((DST) + J * 4 * V2_A_ROW) = t_;                        \
  }
#define V2_STORE_X(P, BUF)                                                           \
  {                                                                                  \
    unsigned char* dst_ = lds + V2_OFF_A + (BUF) * V2_A_TILE + lds_lane;             \
    V2_STORE_X1(P, 0, dst_) V2_STORE_X1(P, 1, dst_) V2_STORE_X1(P, 2, dst_) V2_STORE_X1(P, 3, dst_) \
    V2_STORE_X1(P, 4, dst_) V2_STORE_X1(P, 5, dst_) V2_STORE_X1(P, 6, dst_) V2_STORE_X1(P, 7, dst_) \
  }
    // Three rotating register sets: a tile is loaded THREE units before it is stored (the x
    // loads queue behind the HBM weight stream in the CU's memory pipeline, so their latency
    // is HBM-like even though x is L2-resident), and stored two units before it is consumed.
    V2_DECL_X(xa);
    V2_DECL_X(xb);
    V2_DECL_X(xc);
    V2_LOAD_X(xa)  // unit ub
    V2_LOAD_X(xb)  // unit ub + 1
    V2_LOAD_X(xc)  // unit ub + 2
    V2_STORE_X(xa, 0)
    V2_STORE_X(xb, 1)
    V2_LOAD_X(xa)  // unit ub + 3
    V2_LOAD_X(xb)  // unit ub + 4
    __syncthreads();
    const unsigned long long tpro = dbgp ? __builtin_amdgcn_s_memtime() : 0;
    int wbuf = 2;
    int tile = tile0, chunk = chunk0;
    int u = ub;
#define V2_PRODUCER_STEP(P)                                 \
  V2_STORE_X(P, wbuf) /* unit u + 2 */                      \
  V2_LOAD_X(P)        /* unit u + 5 */                      \
  wbuf = wbuf == V2_XR - 1 ? 0 : wbuf + 1;                  \
  V2_TBAR()                                                 \
  if (chunk == chunks - 1 || u + 1 >= ue) {                 \
    __syncthreads(); /* the consumers' k-half reduction */  \
    chunk = 0;                                              \
    ++tile;                                                 \
  } else {                                                  \
    ++chunk;                                                \
  }                                                         \
  ++u;                                                      \
  if (u >= ue) break;
    for (;;) {
      V2_PRODUCER_STEP(xc)
      V2_PRODUCER_STEP(xa)
      V2_PRODUCER_STEP(xb)
    }
#undef V2_PRODUCER_STEP
    if (dbgp) { dbgp[0] = tpro - tstart; dbgp[1] = __builtin_amdgcn_s_memtime() - tpro; dbgp[2] = bwait; dbgp[3] = ue - ub; }
    return;
  }

  if (wv >= 8) {
    // ================================== weight loaders ================================== //
    const int half = wv - 8;                    // rows half*64 .. +63
    const int piece = lane & 3, rsub = lane >> 2;  // 4 lanes x 16 B = the unit's 64-B row slice
    int lu = ub, ltile = tile0, lchunk = chunk0;
    const unsigned char* wrow[4];
    int64_t srow;  // scale row of this lane (row half*64 + lane)
    auto set_tile = [&](int t) {
      const int nb = t % p.nblocks;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int64_t r = (int64_t)nb * V2_BN + half * 64 + q * 16 + rsub;
        if (r >= p.n) r = p.n - 1;
        wrow[q] = (const unsigned char*)p.w + r * p.w_stride * 4 + piece * 16;
      }
      srow = (int64_t)nb * V2_BN + half * 64 + lane;
      if (srow >= p.n) srow = p.n - 1;
    };
    set_tile(ltile);
    auto advance = [&]() {
      if (lu + 1 < ue) {
        ++lu;
        if (++lchunk == chunks) {
          lchunk = 0;
          ++ltile;
          set_tile(ltile);
        }
      }
    };
    // NOTE: plain scalars + macros on purpose -- structs/arrays passed through lambdas ended up
    // in scratch memory (hipcc did not promote them to registers).
#define V2_DECL_W(P) i32x4 P##0, P##1, P##2, P##3; float P##s, P##z
#define V2_LOAD_W(P)                                                        \
  if (!(p.ablate & 2)) {                                                    \
    const int kb_ = lchunk * 64; /* byte offset of the unit in the row */   \
    P##0 = *reinterpret_cast<const i32x4*>(wrow[0] + kb_);                     \
    P##1 = *reinterpret_cast<const i32x4*>(wrow[1] + kb_);                     \
    P##2 = *reinterpret_cast<const i32x4*>(wrow[2] + kb_);                     \
    P##3 = *reinterpret_cast<const i32x4*>(wrow[3] + kb_);                     \
    const int gi_ = group_of(lchunk);                                       \
    P##s = p.scales[srow * p.s_stride + gi_];                               \
    P##z = p.zeros[srow * p.s_stride + gi_];                                \
    advance();                                                              \
  } else { advance(); }
#define V2_STORE_W(P, SLOT)                                                               \
  {                                                                                       \
    unsigned char* wt_ = lds + V2_OFF_W + (SLOT) * 8192;                                  \
    *reinterpret_cast<i32x4*>(wt_ + st_off[0]) = P##0;                                       \
    *reinterpret_cast<i32x4*>(wt_ + st_off[1]) = P##1;                                       \
    *reinterpret_cast<i32x4*>(wt_ + st_off[2]) = P##2;                                       \
    *reinterpret_cast<i32x4*>(wt_ + st_off[3]) = P##3;                                       \
    uint2 sz_; /* (s, -z*s) as packed fp16 pairs: one fp32 product, one rounding */       \
    sz_.x = v2_bcast(P##s);                                                               \
    sz_.y = v2_bcast(-P##z * P##s);                                                       \
    *reinterpret_cast<uint2*>(lds + V2_OFF_S + (SLOT) * 1024 + (half * 64 + lane) * 8) = sz_; \
  }
    int st_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) st_off[q] = v2_wslot(half * 64 + q * 16 + rsub, piece);
    // prologue: units ub .. ub+D+PF-1 in one round trip
    V2_DECL_W(t0);
    V2_DECL_W(t1);
    V2_DECL_W(w0);
    V2_DECL_W(w1);
    V2_DECL_W(w2);
    V2_DECL_W(w3);
    V2_DECL_W(w4);
    V2_DECL_W(w5);
    V2_LOAD_W(t0)
    V2_LOAD_W(t1)
    V2_LOAD_W(w0)
    V2_LOAD_W(w1)
    V2_LOAD_W(w2)
    V2_LOAD_W(w3)
    V2_LOAD_W(w4)
    V2_LOAD_W(w5)
    V2_STORE_W(t0, 0)
    V2_STORE_W(t1, 1)
    __syncthreads();
    const unsigned long long tpro = dbgp ? __builtin_amdgcn_s_memtime() : 0;
    int wslot = V2_D % V2_RW;
    int tile = tile0, chunk = chunk0;
    int u = ub;
#define V2_LOADER_STEP(P)                                   \
  V2_STORE_W(P, wslot) /* unit u + D */                     \
  V2_LOAD_W(P)         /* unit u + D + PF */                \
  wslot = wslot == V2_RW - 1 ? 0 : wslot + 1;               \
  V2_TBAR()                                                 \
  if (chunk == chunks - 1 || u + 1 >= ue) {                 \
    __syncthreads();                                        \
    chunk = 0;                                              \
    ++tile;                                                 \
  } else {                                                  \
    ++chunk;                                                \
  }                                                         \
  ++u;                                                      \
  if (u >= ue) break;
    for (;;) {
      V2_LOADER_STEP(w0)
      V2_LOADER_STEP(w1)
      V2_LOADER_STEP(w2)
      V2_LOADER_STEP(w3)
      V2_LOADER_STEP(w4)
      V2_LOADER_STEP(w5)
    }
    if (dbgp) { dbgp[0] = tpro - tstart; dbgp[1] = __builtin_amdgcn_s_memtime() - tpro; dbgp[2] = bwait; dbgp[3] = ue - ub; }
#undef V2_LOADER_STEP
#undef V2_STORE_W
#undef V2_LOAD_W
#undef V2_DECL_W
    return;
  }

  // ===================================== consumers ====================================== //
  const int ng = wv & 3, kh = wv >> 2;
  const int nl = lane & 31, h = lane >> 5;
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));

  f32x16 acc[MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  };
  zero_acc();

  // per-lane LDS offsets
  const int wrow_l = ng * 32 + nl;
  const int wpiece = kh * 2 + h;  // 16-B piece (4 words) of the row slice this lane dequantises
  const int woff = v2_wslot(wrow_l, wpiece);
  const int soff = wrow_l * 8;
  const int aoff = nl * V2_A_ROW + kh * 128 + h * 64;

  auto flush = [&](int t, int c_lo, int c_hi) {
    const int mblk = t / p.nblocks, nb = t - mblk * p.nblocks;
    const int64_t m0 = (int64_t)mblk * V2_BM;
    const bool full = (c_lo == 0 && c_hi == chunks - 1);
    if (!full) {
      const int w0 = (int)(((int64_t)t * chunks) / p.upw);
      const int slot = (int)blockIdx.x - w0;
      float* ws = p.workspace + (((int64_t)t * p.slots + slot) * 4 + ng) * (V2_SLAB / 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
          float* dst = ws + ((mt * 4 + g) * 64 + lane) * 4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int old = 0;
      if (lane == 0)
        old = __hip_atomic_fetch_add(&p.counters[t * 4 + ng], c_hi - c_lo + 1, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old + (c_hi - c_lo + 1) != chunks) return;
      if (lane == 0)
        __hip_atomic_store(&p.counters[t * 4 + ng], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      zero_acc();
      const int w1 = (int)((((int64_t)t + 1) * chunks - 1) / p.upw);
      for (int sl = 0; sl <= w1 - w0; ++sl) {
        const float* wr = p.workspace + (((int64_t)t * p.slots + sl) * 4 + ng) * (V2_SLAB / 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + ((mt * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] += v[e];
          }
      }
    }
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t mrow = m0 + nl + mt * 32;
      if (mrow >= p.m) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t nn = (int64_t)nb * V2_BN + ng * 32 + 8 * g + 4 * h;
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t nc = (nn + e) < p.n ? (nn + e) : (p.n - 1);
          float v = acc[mt][4 * g + e];
          if (has_bias) v += f16_bits_to_f32(p.bias[nc]);
          o[e] = f32_to_f16_bits(v);
        }
        if (nn + 3 < p.n && (p.n & 3) == 0) {
          uint2 pk;
          pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
          *reinterpret_cast<uint2*>(p.out + mrow * p.n + nn) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nn + e < p.n) p.out[mrow * p.n + nn + e] = o[e];
        }
      }
    }
  };

  __syncthreads();  // prologue barrier: units ub, ub+1 staged
  const unsigned long long tpro = dbgp ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long tflush = 0;
  int tile = tile0, chunk = chunk0, seg_lo = chunk0;
  int rbuf = 0, wslot = 0;
  for (int u = ub; u < ue; ++u) {
    if (!(p.ablate & 4)) {
    const unsigned char* wt = lds + V2_OFF_W + wslot * 8192;
    const Q4 wq = *reinterpret_cast<const Q4*>(wt + woff);
    const uint2 sz = *reinterpret_cast<const uint2*>(lds + V2_OFF_S + wslot * 1024 + soff);
    const unsigned char* ab = lds + V2_OFF_A + rbuf * V2_A_TILE + aoff;
    f16x8 acur[MT], anxt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acur[mt] = *reinterpret_cast<const f16x8*>(ab + mt * 32 * V2_A_ROW);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          anxt[mt] = *reinterpret_cast<const f16x8*>(ab + mt * 32 * V2_A_ROW + (s + 1) * 16);
      }
      const uint32_t word = s == 0 ? wq.x : s == 1 ? wq.y : s == 2 ? wq.z : wq.w;
      const Q4 wf = v2_dequant(word, sz.x, sz.y, magic);
      const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, acur[mt], acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acur[mt] = anxt[mt];
    }
    }
    rbuf = rbuf == V2_XR - 1 ? 0 : rbuf + 1;
    wslot = wslot == V2_RW - 1 ? 0 : wslot + 1;
    V2_TBAR()
    if (chunk == chunks - 1 || u + 1 >= ue) {
      // sum the two k-halves through LDS, then the kh = 0 wave flushes the tile segment
      float* red = reinterpret_cast<float*>(lds + V2_OFF_R + ng * 8192);
      if (kh == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(red + ((mt * 4 + g) * 64 + lane) * 4) =
                f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
      }
      __syncthreads();
      const unsigned long long f0 = dbgp ? __builtin_amdgcn_s_memtime() : 0;
      if (kh == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((mt * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] += v[e];
          }
        flush(tile, seg_lo, chunk);
      }
      if (dbgp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tflush += __builtin_amdgcn_s_memtime() - f0; }
      zero_acc();
      chunk = 0;
      seg_lo = 0;
      ++tile;
    } else {
      ++chunk;
    }
  }
  if (dbgp) { dbgp[0] = tpro - tstart; dbgp[1] = __builtin_amdgcn_s_memtime() - tpro; dbgp[2] = bwait; dbgp[3] = tflush; }
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct V2Plan {
  int mblocks, nblocks, chunks, total_units, upw, grid, slots;
};

static int v2_num_cus() {
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

static V2Plan v2_plan(int64_t m, int64_t n, int64_t k) {
  V2Plan pl;
  pl.mblocks = (int)((m + V2_BM - 1) / V2_BM);
  pl.nblocks = (int)((n + V2_BN - 1) / V2_BN);
  pl.chunks = (int)((k + V2_CK - 1) / V2_CK);
  pl.total_units = pl.mblocks * pl.nblocks * pl.chunks;
  int target = v2_num_cus();  // one persistent 12-wave workgroup per CU
  if (const char* e = getenv("LL_GEMM2_WGS")) {
    const int v = atoi(e);
    if (v > 0) target = v;
  }
  int upw = (pl.total_units + target - 1) / target;
  const int min_upw = (pl.chunks + (V2_MAX_SLOTS - 2) - 1) / (V2_MAX_SLOTS - 2);
  if (upw < min_upw) upw = min_upw;
  if (upw < 2) upw = 2;
  pl.upw = upw;
  pl.grid = (pl.total_units + upw - 1) / upw;
  int slots = (pl.chunks - 1) / upw + 2;
  if (slots > V2_MAX_SLOTS) slots = V2_MAX_SLOTS;
  pl.slots = slots;
  return pl;
}

// exported for gemm_wq.hip's dispatcher
extern "C" int ll_w4a16_v2_workspace(int64_t m, int64_t n, int64_t k, int64_t* floats, int64_t* ints) {
  const V2Plan pl = v2_plan(m, n, k);
  const int64_t tiles = (int64_t)pl.mblocks * pl.nblocks;
  if (floats) *floats = tiles * pl.slots * V2_SLAB;
  if (ints) *ints = tiles * 4;
  return LL_OK;
}

extern "C" int ll_w4a16_v2_supported(int64_t m, int64_t n, int64_t k, int group_size) {
  if (getenv("LL_GEMM_V1")) return 0;
  return (group_size % 128 == 0) && (k % 128 == 0) && (n >= 1) && (m >= 1);
}

extern "C" int ll_w4a16_v2_launch(void* out, const void* x, const int32_t* qweight, const float* scales,
                                  const float* zeros, const void* bias, int64_t m, int64_t n, int64_t k,
                                  int group_size, int64_t x_stride_m, int64_t qw_stride_n, int64_t s_stride_n,
                                  float* workspace, int32_t* counters, void* stream) {
  const V2Plan pl = v2_plan(m, n, k);
  V2Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.w = (const uint32_t*)qweight; p.scales = scales;
  p.zeros = zeros; p.bias = (const uint16_t*)bias; p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m; p.w_stride = qw_stride_n; p.s_stride = s_stride_n;
  p.nblocks = pl.nblocks; p.chunks = pl.chunks; p.total_units = pl.total_units; p.upw = pl.upw; p.slots = pl.slots;
  p.gdiv = group_size / 128;
  p.gshift = -1;
  if ((p.gdiv & (p.gdiv - 1)) == 0) {
    int sh = 0;
    while ((1 << sh) < p.gdiv) ++sh;
    p.gshift = sh;
  }
  p.dbg = nullptr;
  p.ablate = getenv("LL_GEMM2_ABLATE") ? atoi(getenv("LL_GEMM2_ABLATE")) : 0;
  if (const char* e = getenv("LL_GEMM_TRACE")) p.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)wgemm2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);
    hipFuncSetAttribute((const void*)wgemm2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (m <= 32)
    wgemm2_kernel<1><<<dim3((unsigned)pl.grid), V2_THREADS, V2_LDS_BYTES, st>>>(p);
  else
    wgemm2_kernel<2><<<dim3((unsigned)pl.grid), V2_THREADS, V2_LDS_BYTES, st>>>(p);
  return LL_LAUNCH_CHECK();
}
