// W4A16 dequant-GEMM, decode engine (M <= 64, group_size % 128 == 0: the AWQ/GPTQ g128 case).
// Reference semantics: lite_llama/kernels/quantization/w4a16.py:28-207 (see gemm_wq.hip for the
// generic engine that serves every other shape / group size and the 8-bit formats).
//
// What the round-1 measurements (benchmarks/gemm_shapes.py + ablations) said, and what
// this engine does about it:
//   * lane-per-row weight loads stream at 3.7 TB/s, 64-B-per-row coalesced loads at 4.3+ TB/s
//     -> dedicated LOADER waves fetch the [128 rows x 64 B] unit tile with 4 lanes per row and stage
//        it (XOR-swizzled, conflict-free) in an LDS ring; scale/zero pairs ride along;
//   * one consumer wave per SIMD is issue-bound -> 8 consumer waves (2 per SIMD): 4 row groups x
//     2 k-halves of every unit, each doing 4 MFMA steps; the k-halves are summed through LDS once
//     per tile segment;
//   * every role is a single instruction stream (~5 cycles per instruction): the per-unit cost of
//     the WHOLE workgroup is the longest role's instruction count, so loaders/producers carry no
//     address arithmetic in the loop (fixed per-lane VGPR offsets + one scalar base per unit) and
//     the unit sequence is advanced with branch-free scalar selects.  No control flow ever
//     surrounds a global load (hipcc's waitcnt pass drains vmcnt(0) otherwise);
//   * tile merges: a release fence or a returning ticket atomic in the middle of the stream stalls
//     the workgroup for microseconds -> STATIC ownership.  Each workgroup runs its range
//     tail-segment FIRST and head-segment LAST: the tail (first chunks of a tile finished by the
//     next workgroup) is parked in a slab with write-through stores and flagged one unit later
//     (fire and forget); the head (last chunks of a tile) is finished at the very end by its
//     owner, which finds the earlier contributors' flags already set, adds their slabs and writes
//     the output.  Waits only ever point at lower-numbered workgroups (dispatched earlier).
// Roles (wave id): 0-7 consumers (ng = w & 3, kh = w >> 2), 8-9 weight loaders (64 rows each),
// 10-11 activation producers (32 rows each).  One s_barrier per unit for everybody.
#include <stdlib.h>

#include "common.h"

#define V2_BN 128
#define V2_BM 64
#define V2_CK 128
#define V2_THREADS 768
#define V2_SLAB (V2_BN * V2_BM)
#define V2_MAX_SLOTS 12
#define V2_RW 4  // LDS weight ring depth (units)
#define V2_D 2   // a unit is staged in LDS this many iterations before it is consumed

struct alignas(16) Q4 {
  uint32_t x, y, z, w;
};

struct V2Params {
  uint16_t* out;
  const uint16_t* x;
  const uint32_t* w;
  const float* scales;
  const float* zeros;
  const void* packed;  // optional [K/g][N] x 8 B (s, -z*s) fp16 pairs, see ll_w4a16_pack_scales
  const uint16_t* bias;
  float* workspace;
  int32_t* counters;
  int64_t m, n, k;
  int64_t x_stride, w_stride, s_stride;  // elements / int32 words / floats per row
  int nblocks, chunks, total_units, upw, slots;
  // tile-group split (gt > 0, see v2_plan): gt workgroups per tile; contributor j owns
  // gbase + (j < grem) chunks, the last one (the owner) gbase + glead
  int gt, gbase, grem, glead;
  int gshift;  // log2(group_size / 128) when a power of two, else -1
  int gdiv;    // group_size / 128
  int epi;     // 0: out[m, n];  1: rows are (gate_j, up_j) pairs -> out[m, n/2] = swiglu (ll_w4a16_gateup_swiglu)
#ifdef V2_TIMELINE
  int tlwave;                    // debug: which consumer wave (0..7) writes the role-0 stamps
  unsigned long long* timeline;  // debug: [workgroup][role 3][V2_TLN] s_memrealtime stamps (benchmarks/gemm_timeline.py)
#endif
};

// debug timeline (-DV2_TIMELINE): role 0 consumer wave 0, 1 loader wave 8, 2 producer wave 10;
// slot 0 kernel entry, 1 prologue barrier passed, then per unit v < V2_TLU: 2+3v data ready (loader:
// after the LDS store, i.e. after the wait for the loaded data), 3+3v arrived at the unit barrier,
// 4+3v left it; last slot: role done
#define V2_TLU 40
#define V2_TLN (6 + 3 * V2_TLU)  // + 3 prologue stamps per role: table built, loads issued, first data stored
#ifdef V2_TIMELINE
#define V2_TL(ROLE, IDX)                                                                                       \
  if (p.timeline && lane == 0)                                                                                 \
    p.timeline[((size_t)blockIdx.x * 3 + (ROLE)) * V2_TLN + (IDX)] = __builtin_amdgcn_s_memrealtime();
#define V2_TLV(ROLE, V, K) if ((V) < V2_TLU) V2_TL(ROLE, 2 + 3 * (V) + (K))
#define V2_TLW (wv == p.tlwave)  // the consumer wave that writes the role-0 stamps (LL_GEMM_TL_WAVE)
#else
#define V2_TL(ROLE, IDX)
#define V2_TLV(ROLE, V, K)
#define V2_TLW false
#endif

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
#define V2_A_TILE (V2_BM * V2_A_ROW)        // 17408
#define V2_XR 4                             // LDS activation ring depth (units)
#define V2_OFF_A 0                          // V2_XR x-tiles
#define V2_OFF_W (V2_XR * V2_A_TILE)        // RW weight tiles of 8192 B
#define V2_OFF_S (V2_OFF_W + V2_RW * 8192)  // RW scale tiles: 128 rows x (s, -z*s) fp16 pairs x2 = 8 B
#define V2_OFF_R (V2_OFF_S + V2_RW * 1024)  // reduce buffer: 4 row groups x 8 KB
#define V2_OFF_T (V2_OFF_R + 4 * 8192)      // unit table: 16 B per unit of the workgroup
#define V2_MAX_UNITS 1024
#define V2_LDS_BYTES (V2_OFF_T + V2_MAX_UNITS * 16)

// swizzled byte offset of 16-B piece p (0..3) of row r (0..127) inside an 8-KB weight tile
__device__ __forceinline__ int v2_wslot(int r, int p) { return (r * 4 + (p ^ ((r >> 2) & 3))) * 16; }

// The unit sequence of one workgroup (all scalar).  Global unit g = tile * chunks + chunk; the
// workgroup owns [ub, ue).  Order: tail segment (first chunks of the last tile), full tiles, head
// segment (last chunks of the first tile).  LT / NF / LH = units in each part.

// ABL (debug builds only, -DV2_DEBUG_ABLATE): 1 no x loads, 2 no weight loads, 4 no compute, 8 no flush
template <int MT, int PK, int ABL>
__global__ __launch_bounds__(V2_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgemm2_kernel(const V2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = p.chunks;
  int ub, ue;
  if (p.gt) {
    const int gtile = (int)blockIdx.x / p.gt, j = (int)blockIdx.x - gtile * p.gt;
    const int lo = j * p.gbase + (j < p.grem ? j : p.grem);
    ub = gtile * chunks + lo;
    ue = j == p.gt - 1 ? (gtile + 1) * chunks : ub + p.gbase + (j < p.grem ? 1 : 0);
  } else {
    ub = blockIdx.x * p.upw;
    ue = ub + p.upw;
    if (ue > p.total_units) ue = p.total_units;
  }
  if (ub >= ue) return;
  const int cnt = ue - ub;
  if (V2_TLW) { V2_TL(0, 0) }
  if (wv == 8) { V2_TL(1, 0) }
  if (wv == 10) { V2_TL(2, 0) }
  const int tA = ub / chunks, cA = ub - tA * chunks;
  const int tZ = (ue - 1) / chunks, cZ = (ue - 1) - tZ * chunks;
  int LT = 0, LH = cnt;  // a range inside one tile runs as a single "head" segment
  if (tA != tZ) {
    LT = (cZ != chunks - 1) ? cZ + 1 : 0;
    LH = (cA != 0) ? chunks - cA : 0;
  }
  const int NF = cnt - LT - LH;
  const int tF = tA + ((LH > 0 && tA != tZ) ? 1 : 0);

  // The unit table: one 16-B entry per unit of this workgroup, in execution order, built once by
  // all threads.  Every role then gets the offsets of "its" unit with one broadcast ds_read_b128
  // instead of carrying the (tile, chunk) arithmetic in its instruction stream.
  //   .x weight byte offset of (tile row 0, chunk)   .y scale byte offset of (tile row 0, group)
  //   .z activation byte offset of the chunk          .w seg_end | chunk << 1 | tile << 13
  i32x4* tab = reinterpret_cast<i32x4*>(lds + V2_OFF_T);
  for (int v = tid; v < cnt; v += V2_THREADS) {
    int t, c;
    if (v < LT) {
      t = tZ;
      c = v;
    } else if (v < LT + NF) {
      const int q = v - LT;
      const int dq = q / chunks;
      t = tF + dq;
      c = q - dq * chunks;
    } else {
      t = tA;
      c = cA + (v - LT - NF);
    }
    const int se = (c == chunks - 1) | (v == LT - 1) | (v == cnt - 1);
    i32x4 e;
    e.x = (int)((uint32_t)t * (uint32_t)(V2_BN * 4) * (uint32_t)p.w_stride + (uint32_t)c * 64u);
    if constexpr (PK)
      e.y = (int)(((uint32_t)(c >> p.gshift) * (uint32_t)p.n + (uint32_t)t * V2_BN) * 8u);
    else
      e.y = (int)((uint32_t)t * (uint32_t)(V2_BN * 4) * (uint32_t)p.s_stride + (uint32_t)(c >> p.gshift) * 4u);
    e.z = c * (V2_CK * 2);
    e.w = se | (c << 1) | (t << 13);
    tab[v] = e;
  }
  __syncthreads();
  if (V2_TLW) { V2_TL(0, 3 + 3 * V2_TLU) }
  if (wv == 8) { V2_TL(1, 3 + 3 * V2_TLU) }
  if (wv == 10) { V2_TL(2, 3 + 3 * V2_TLU) }
  const int last = cnt - 1;
#define V2_ENTRY(V) tab[(V) < last ? (V) : last]  // sticks on the last unit past the end

  if (wv >= 10) {
    // =============================== activation producers =============================== //
    // rows half*32 .. +31 of the [64 x 128] fp16 x-tile; lane = (row sub 0..3, 16-B column 0..15)
    // The memory roles have few instructions per unit but every one of them gates the stream:
    // they outrank the consumers sharing their SIMD.
    __builtin_amdgcn_s_setprio(3);
    const int half = wv - 10;
    const int xcol = lane & 15, xrsub = lane >> 4;
    uint32_t roff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = half * 32 + j * 4 + xrsub;
      const int64_t rc = row < p.m ? row : p.m - 1;  // rows >= M feed only unstored outputs
      roff[j] = (uint32_t)(rc * p.x_stride * 2 + xcol * 16);
    }
    const unsigned char* xbase = (const unsigned char*)p.x;
    const uint32_t lds_lane = (uint32_t)((half * 32 + xrsub) * V2_A_ROW + xcol * 16);
    int lv = 0;  // load position
    int xo = V2_ENTRY(0).z;
#define V2_DECL_X(P) i32x4 P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7
#define V2_LOAD_X1(P, J) if constexpr (!(ABL & 1)) P##J = *reinterpret_cast<const i32x4*>(xbase + (roff[J] + (uint32_t)xo));
#define V2_LOAD_X(P)                                                    \
  {                                                                     \
    V2_LOAD_X1(P, 0) V2_LOAD_X1(P, 1) V2_LOAD_X1(P, 2) V2_LOAD_X1(P, 3) \
    V2_LOAD_X1(P, 4) V2_LOAD_X1(P, 5) V2_LOAD_X1(P, 6) V2_LOAD_X1(P, 7) \
    ++lv;                                                               \
    xo = V2_ENTRY(lv).z;                                                \
  }
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
    // Three rotating register sets: a tile is loaded three units before it is stored and stored
    // two units before it is consumed (the x loads queue behind the HBM weight stream in the
    // CU's memory pipeline, so their latency is HBM-like even though x is L2-resident).
    V2_DECL_X(xa);
    V2_DECL_X(xb);
    V2_DECL_X(xc);
    V2_LOAD_X(xa)  // unit 0
    V2_LOAD_X(xb)  // unit 1
    V2_LOAD_X(xc)  // unit 2
    if (wv == 10) { V2_TL(2, 4 + 3 * V2_TLU) }
    V2_STORE_X(xa, 0)
    V2_STORE_X(xb, 1)
    if (wv == 10) { V2_TL(2, 5 + 3 * V2_TLU) }
    V2_LOAD_X(xa)  // unit 3
    V2_LOAD_X(xb)  // unit 4
    __syncthreads();
    if (wv == 10) { V2_TL(2, 1) }
    int wbuf = 2;
    int cv = 0;  // the unit the consumers are on (barrier schedule)
#define V2_PRODUCER_STEP(P)                                                   \
  {                                                                           \
    const int fl_ = __builtin_amdgcn_readfirstlane(tab[cv].w);                \
    V2_STORE_X(P, wbuf) /* unit v + 2 */                                      \
    if (wv == 10) { V2_TLV(2, cv, 0) }                                        \
    V2_LOAD_X(P)        /* unit v + 5 */                                      \
    wbuf = wbuf == V2_XR - 1 ? 0 : wbuf + 1;                                  \
    if (wv == 10) { V2_TLV(2, cv, 1) }                                        \
    __syncthreads();                                                          \
    if (wv == 10) { V2_TLV(2, cv, 2) }                                        \
    if (fl_ & 1) __syncthreads(); /* the consumers' k-half reduction */       \
    ++cv;                                                                     \
  }
    // full rotations in a single-exit loop (a break between steps made hipcc drain vmcnt(0)),
    // then the 0-2 leftover steps
    for (int it = cnt / 3; it > 0; --it) {
      V2_PRODUCER_STEP(xc)
      V2_PRODUCER_STEP(xa)
      V2_PRODUCER_STEP(xb)
    }
    if (cv < cnt) V2_PRODUCER_STEP(xc)
    if (cv < cnt) V2_PRODUCER_STEP(xa)
#undef V2_PRODUCER_STEP
    return;
  }

  if (wv >= 8) {
    // ================================== weight loaders ================================== //
    __builtin_amdgcn_s_setprio(3);
    const int half = wv - 8;                       // rows half*64 .. +63
    const int piece = lane & 3, rsub = lane >> 2;  // 4 lanes x 16 B = the unit's 64-B row slice
    uint32_t woff[4];                              // byte offsets relative to (tile row 0, chunk 0)
    int st_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = half * 64 + q * 16 + rsub;
      woff[q] = (uint32_t)((int64_t)r * p.w_stride * 4 + piece * 16);
      st_off[q] = v2_wslot(r, piece);
    }
    const uint32_t soff = PK ? (uint32_t)((half * 64 + lane) * 8) : (uint32_t)((int64_t)(half * 64 + lane) * p.s_stride * 4);
    const unsigned char* pbase = (const unsigned char*)p.packed;
    const unsigned char* wbase = (const unsigned char*)p.w;
    const unsigned char* sbase = (const unsigned char*)p.scales;
    const unsigned char* zbase = (const unsigned char*)p.zeros;
    int lv = 0;  // load position
    i32x4 le = V2_ENTRY(0);
    // NOTE: plain scalars + macros on purpose -- structs/arrays passed through lambdas ended up
    // in scratch memory (hipcc did not promote them to registers).
#define V2_DECL_W(P) i32x4 P##0, P##1, P##2, P##3; float P##s, P##z; uint2 P##p
#define V2_LOAD_W(P)                                                                     \
  {                                                                                      \
    const uint32_t wo_ = (uint32_t)le.x, so_ = (uint32_t)le.y + soff;                    \
    if constexpr (!(ABL & 2)) {                                                          \
    P##0 = *reinterpret_cast<const i32x4*>(wbase + (wo_ + woff[0]));                     \
    P##1 = *reinterpret_cast<const i32x4*>(wbase + (wo_ + woff[1]));                     \
    P##2 = *reinterpret_cast<const i32x4*>(wbase + (wo_ + woff[2]));                     \
    P##3 = *reinterpret_cast<const i32x4*>(wbase + (wo_ + woff[3]));                     \
    if constexpr (PK) {                                                                  \
      P##p = *reinterpret_cast<const uint2*>(pbase + so_);                               \
    } else {                                                                             \
      P##s = *reinterpret_cast<const float*>(sbase + so_);                               \
      P##z = *reinterpret_cast<const float*>(zbase + so_);                               \
    }                                                                                    \
    }                                                                                    \
    ++lv;                                                                                \
    le = V2_ENTRY(lv);                                                                   \
  }
#define V2_STORE_W(P, SLOT)                                                               \
  {                                                                                       \
    unsigned char* wt_ = lds + V2_OFF_W + (SLOT) * 8192;                                  \
    *reinterpret_cast<i32x4*>(wt_ + st_off[0]) = P##0;                                    \
    *reinterpret_cast<i32x4*>(wt_ + st_off[1]) = P##1;                                    \
    *reinterpret_cast<i32x4*>(wt_ + st_off[2]) = P##2;                                    \
    *reinterpret_cast<i32x4*>(wt_ + st_off[3]) = P##3;                                    \
    uint2 sz_; /* (s, -z*s) as packed fp16 pairs: one fp32 product, one rounding */       \
    if constexpr (PK) {                                                                   \
      sz_ = P##p;                                                                         \
    } else {                                                                              \
      sz_.x = v2_bcast(P##s);                                                             \
      sz_.y = v2_bcast(-P##z * P##s);                                                     \
    }                                                                                     \
    *reinterpret_cast<uint2*>(lds + V2_OFF_S + (SLOT) * 1024 + (half * 64 + lane) * 8) = sz_; \
  }
    // prologue: units 0 .. D+PF-1 in one round trip
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
    if (wv == 8) { V2_TL(1, 4 + 3 * V2_TLU) }
    V2_STORE_W(t0, 0)
    V2_STORE_W(t1, 1)
    if (wv == 8) { V2_TL(1, 5 + 3 * V2_TLU) }
    __syncthreads();
    if (wv == 8) { V2_TL(1, 1) }
    int wslot = V2_D % V2_RW;
    int cv = 0;
#define V2_LOADER_STEP(P)                                                     \
  {                                                                           \
    const int fl_ = __builtin_amdgcn_readfirstlane(tab[cv].w);                \
    V2_STORE_W(P, wslot) /* unit v + D */                                     \
    if (wv == 8) { V2_TLV(1, cv, 0) }                                         \
    V2_LOAD_W(P)         /* unit v + D + PF */                                \
    wslot = wslot == V2_RW - 1 ? 0 : wslot + 1;                               \
    if (wv == 8) { V2_TLV(1, cv, 1) }                                         \
    __syncthreads();                                                          \
    if (wv == 8) { V2_TLV(1, cv, 2) }                                         \
    if (fl_ & 1) __syncthreads();                                             \
    ++cv;                                                                     \
  }
    for (int it = cnt / 6; it > 0; --it) {
      V2_LOADER_STEP(w0)
      V2_LOADER_STEP(w1)
      V2_LOADER_STEP(w2)
      V2_LOADER_STEP(w3)
      V2_LOADER_STEP(w4)
      V2_LOADER_STEP(w5)
    }
    if (cv < cnt) V2_LOADER_STEP(w0)
    if (cv < cnt) V2_LOADER_STEP(w1)
    if (cv < cnt) V2_LOADER_STEP(w2)
    if (cv < cnt) V2_LOADER_STEP(w3)
    if (cv < cnt) V2_LOADER_STEP(w4)
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

  // contribution flag of a parked partial, posted once its write-through stores have landed
  int32_t* pend_ctr = nullptr;
  int pend_val = 0;
  auto post_pending = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(pend_ctr, pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_ctr = nullptr;
  };

  auto flush = [&](int t, int c_lo, int c_hi) {
    const int w0 = p.gt ? t * p.gt : (int)(((int64_t)t * chunks) / p.upw);  // first contributor of the tile
    const int slot = (int)blockIdx.x - w0;
    if (c_hi != chunks - 1) {
      // contributor: park the partial in this workgroup's slab (flag follows, see post_pending)
      float* ws = p.workspace + (((int64_t)t * p.slots + slot) * 4 + ng) * (V2_SLAB / 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
          float* dst = ws + ((mt * 4 + g) * 64 + lane) * 4;
          if constexpr (!(ABL & 16))
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
      pend_ctr = &p.counters[t * 4 + ng];
      pend_val = c_hi - c_lo + 1;
      return;
    }
    if (c_lo != 0 && !(ABL & 32)) {
      // owner: chunks [0, c_lo) were computed by the `slot` earlier contributors
      int32_t* ctr = &p.counters[t * 4 + ng];
      for (;;) {
        const int seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(seen) == c_lo) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // The slabs were written through to memory (sc1 stores) before their flag; read them with
      // agent-coherent (sc1) loads instead of paying an acquire fence (buffer_inv sc1, ~4 us when
      // every CU does it at once).  Two slabs in flight; the second of an odd tail is masked to +0.
      for (int sl = 0; sl < slot; sl += 2) {
        const int sl2 = sl + 1 < slot ? sl + 1 : sl;
        const int keep = sl + 1 < slot ? -1 : 0;
        const float* wa = p.workspace + (((int64_t)t * p.slots + sl) * 4 + ng) * (V2_SLAB / 4) + lane * 4;
        const float* wb = p.workspace + (((int64_t)t * p.slots + sl2) * 4 + ng) * (V2_SLAB / 4) + lane * 4;
        i32x4 va[MT * 4], vb[MT * 4];
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(va[i]) : "v"(wa + i * 256) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(vb[i]) : "v"(wb + i * 256) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) asm volatile("" : "+v"(va[i]), "+v"(vb[i]));  // uses stay below the wait
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[mt][4 * g + e] += __int_as_float(va[mt * 4 + g][e]);
              acc[mt][4 * g + e] += __int_as_float(vb[mt * 4 + g][e] & keep);
            }
      }
    }
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t mrow = nl + mt * 32;
      if (mrow >= p.m) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t nn = (int64_t)t * V2_BN + ng * 32 + 8 * g + 4 * h;
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[mt][4 * g + e];
          if (has_bias) v += f16_bits_to_f32(p.bias[nn + e]);
          o[e] = f32_to_f16_bits(v);
        }
        if (p.epi) {
          // weight rows 2j / 2j+1 are gate_j / up_j: both land in this lane.  Same arithmetic as the
          // stand-alone kernels: the two GEMM outputs rounded to fp16, then silu(g) * u in fp32.
          const float g0 = f16_bits_to_f32(o[0]), u0 = f16_bits_to_f32(o[1]);
          const float g1 = f16_bits_to_f32(o[2]), u1 = f16_bits_to_f32(o[3]);
          const uint32_t s0 = f32_to_f16_bits(g0 * ll_sigmoidf(g0) * u0);
          const uint32_t s1 = f32_to_f16_bits(g1 * ll_sigmoidf(g1) * u1);
          *reinterpret_cast<uint32_t*>(p.out + mrow * (p.n >> 1) + (nn >> 1)) = s0 | (s1 << 16);
        } else {
          uint2 pk;
          pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
          *reinterpret_cast<uint2*>(p.out + mrow * p.n + nn) = pk;
        }
      }
    }
  };

  int cv = 0;
  int fl = __builtin_amdgcn_readfirstlane(tab[0].w);
  int seg_lo = (fl >> 1) & 0xFFF;
  int rbuf = 0, wslot = 0;
  __syncthreads();  // prologue barrier: units 0, 1 staged
  if (V2_TLW) { V2_TL(0, 1) }
  // ALL operands of a unit are fetched ONE UNIT AHEAD, across the barrier (a unit is staged in LDS
  // two steps before it is consumed, so unit v+1 is already there while unit v is multiplied).
  // The consumers are the critical role (they wait at the barrier 20 % of the time, the loaders
  // 55 %): with same-unit reads all 8 consumer waves hit the LDS together right after the barrier
  // (80 ds_read_b128 = ~320 LDS cycles, plus the memory roles' stores) while the matrix pipes
  // idle; one unit ahead, the LDS time hides under the MFMAs.  Each step's prefetch is pinned
  // behind that step's MFMAs with a scheduling barrier (hipcc otherwise sinks the reads to their
  // use to save registers).
  Q4 wq = *reinterpret_cast<const Q4*>(lds + V2_OFF_W + woff);
  uint2 sz = *reinterpret_cast<const uint2*>(lds + V2_OFF_S + soff);
  f16x8 af[4][MT];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      af[s][mt] = *reinterpret_cast<const f16x8*>(lds + V2_OFF_A + aoff + mt * 32 * V2_A_ROW + s * 16);
  for (;;) {
    const int rbuf_n = rbuf == V2_XR - 1 ? 0 : rbuf + 1;
    const int wslot_n = wslot == V2_RW - 1 ? 0 : wslot + 1;
    Q4 wq_n = wq;
    uint2 sz_n = sz;
    f16x8 an[4][MT];
    int fl_n = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) an[s][mt] = af[s][mt];
    if constexpr (!(ABL & 4)) {
      const unsigned char* abn = lds + V2_OFF_A + rbuf_n * V2_A_TILE + aoff;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint32_t word = s == 0 ? wq.x : s == 1 ? wq.y : s == 2 ? wq.z : wq.w;
        Q4 wf;
        if constexpr (ABL & 256) {
          wf.x = word; wf.y = word ^ sz.x; wf.z = word ^ sz.y; wf.w = word;  // debug: no dequant
        } else {
          wf = v2_dequant(word, sz.x, sz.y, magic);
        }
        const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
        if constexpr (ABL & 128) {  // debug: no MFMA
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(wfrag), "v"(af[s][mt]));
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, af[s][mt], acc[mt], 0, 0, 0);
        }
        // next unit, same step
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          an[s][mt] = *reinterpret_cast<const f16x8*>(abn + mt * 32 * V2_A_ROW + s * 16);
        if (s == 0) {
          wq_n = *reinterpret_cast<const Q4*>(lds + V2_OFF_W + wslot_n * 8192 + woff);
          sz_n = *reinterpret_cast<const uint2*>(lds + V2_OFF_S + wslot_n * 1024 + soff);
          // only the flags dword: a 16-B read into temporaries the dequant wants back would force
          // a full lgkmcnt(0) right here
          fl_n = reinterpret_cast<const int*>(&tab[cv + 1 < last ? cv + 1 : last])[3];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      fl_n = V2_ENTRY(cv + 1).w;
    }
    rbuf = rbuf_n;
    wslot = wslot_n;
    // bare s_barrier: __syncthreads() would first drain lgkmcnt(0), i.e. wait for the prefetched
    // reads of the next unit; the consumers have no LDS stores of their own to publish here, and the
    // slots they are reading are not rewritten before two more barriers
    if (V2_TLW) { V2_TLV(0, cv, 1) }
    __builtin_amdgcn_s_barrier();
    if (V2_TLW) { V2_TLV(0, cv, 2) }
    if (pend_ctr) post_pending();  // the previous segment's slab stores are a unit old by now
    const bool se = fl & 1;
    if (se) {
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
      if (kh == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((mt * 4 + g) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] += v[e];
          }
        if constexpr (!(ABL & 8)) flush(fl >> 13, seg_lo, (fl >> 1) & 0xFFF);
      }
      zero_acc();
    }
    if (++cv >= cnt) break;
    fl = __builtin_amdgcn_readfirstlane(fl_n);
    if (se) seg_lo = (fl >> 1) & 0xFFF;
    wq = wq_n;
    sz = sz_n;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[s][mt] = an[s][mt];
  }
  if (pend_ctr) post_pending();
  if (V2_TLW) { V2_TL(0, 2 + 3 * V2_TLU) }
}

// ---------------------------------------------------------------------------------- //
// host side
// ---------------------------------------------------------------------------------- //
struct V2Plan {
  int nblocks, chunks, total_units, upw, grid, slots;
  int gt, gbase, grem, glead;
};

static int v2_num_cus() {  // per device
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

struct V2Knobs {  // the environment is read once per process
  int wgs = 0, lead = 4, gt_cap = -1;
  bool v1 = false;
  V2Knobs() {
    if (const char* e = getenv("LL_GEMM2_WGS")) wgs = atoi(e);
    if (const char* e = getenv("LL_GEMM2_LEAD")) lead = atoi(e);
    if (const char* e = getenv("LL_GEMM2_GT")) gt_cap = atoi(e);
  }
};
static const V2Knobs& v2_knobs() {
  static const V2Knobs k;
  return k;
}

static V2Plan v2_plan(int64_t n, int64_t k) {
  V2Plan pl;
  pl.nblocks = (int)(n / V2_BN);
  pl.chunks = (int)(k / V2_CK);
  pl.total_units = pl.nblocks * pl.chunks;
  const V2Knobs& kn = v2_knobs();
  const int target = kn.wgs > 0 ? kn.wgs : v2_num_cus();  // one persistent 12-wave workgroup per CU
  pl.gt = pl.gbase = pl.grem = pl.glead = 0;
  // Few tiles (N <= 16 K at K = 3584): every tile is shared by gt workgroups and, with equal
  // shares, all of them finish together -- the owner then pays the whole merge chain (store ack,
  // flag, poll, slab loads: ~8 us) after its last unit.  Tile-group split: the owner gets `lead`
  // more chunks than the contributors, so their slabs and flags have landed by the time it is
  // done; the chain collapses to one round of slab loads.
  const int lead = kn.lead;  // measured: 4 units (~3.5 us) cover the contributors' store ack + flag
  int gt = pl.nblocks > 0 ? target / pl.nblocks : 0;
  if (gt > V2_MAX_SLOTS) gt = V2_MAX_SLOTS;
  // the owner adds the slabs two per fabric round trip (more contributors, more rounds) and runs
  // lead units longer than everybody else: measured optimum ~ one workgroup per 5 chunks
  // (28 chunks -> 5, 148 chunks -> as many as there are CUs for)
  const int gt_cap = kn.gt_cap >= 0 ? kn.gt_cap : pl.chunks / 5;
  if (gt > gt_cap) gt = gt_cap;
  if (lead > 0 && gt >= 2 && pl.chunks - lead >= gt) {
    pl.gt = gt;
    pl.glead = lead;
    pl.gbase = (pl.chunks - lead) / gt;
    pl.grem = (pl.chunks - lead) % gt;
    pl.upw = pl.gbase + lead;  // the longest range (table capacity check)
    pl.grid = pl.nblocks * gt;
    pl.slots = gt;
    return pl;
  }
  int upw = (pl.total_units + target - 1) / target;
  // a tile has at most (chunks - 2) / upw + 2 contributors
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

static bool v2_shape_ok(int64_t m, int64_t n, int64_t k) {
  if (!((m >= 1) && (m <= V2_BM) && (n >= V2_BN) && (n % V2_BN == 0) && (k >= V2_CK) && (k % V2_CK == 0)))
    return false;
  if (n * k / 2 >= (1ll << 31) || k / V2_CK > 4095 || n / V2_BN >= (1 << 18)) return false;  // 32-bit table fields
  return v2_plan(n, k).upw <= V2_MAX_UNITS;
}

// (s, -z*s) as fp16 pairs, transposed to [group][row]: a unit's 128 rows become 1 KB contiguous
// (the fp32 [N, K/g] grids cost 256 scattered 4-B requests per unit -- twice the weight stream's).
// Same arithmetic as the loader's in-flight conversion: fp32 product, one rounding each.
__global__ void w4a16_pack_scales_kernel(uint2* packed, const float* scales, const float* zeros, int64_t n,
                                         int64_t groups, int64_t s_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * groups) return;
  const int64_t g = i / n, r = i - g * n;
  const float s = scales[r * s_stride + g], z = zeros[r * s_stride + g];
  uint2 o;
  o.x = v2_bcast(s);
  o.y = v2_bcast(-z * s);
  packed[i] = o;
}

extern "C" int ll_w4a16_pack_scales(void* packed, const float* scales, const float* zeros, int64_t n,
                                    int64_t groups, int64_t s_stride_n, void* stream) {
  if (n <= 0 || groups <= 0) return LL_ERR_SHAPE;
  if (!packed || !scales || !zeros) return LL_ERR_ARG;
  const int64_t total = n * groups;
  w4a16_pack_scales_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      (uint2*)packed, scales, zeros, n, groups, s_stride_n);
  return LL_LAUNCH_CHECK();
}

// exported for gemm_wq.hip's dispatcher
extern "C" int ll_w4a16_v2_supported(int64_t m, int64_t n, int64_t k, int group_size) {
  static const bool v1_forced = getenv("LL_GEMM_V1") != nullptr;  // debug knob, read once
  if (v1_forced) return 0;
  const int gdiv = group_size / 128;
  return v2_shape_ok(m, n, k) && (group_size % 128 == 0) && ((gdiv & (gdiv - 1)) == 0);
}

extern "C" int ll_w4a16_v2_workspace(int64_t m, int64_t n, int64_t k, int64_t* floats, int64_t* ints) {
  if (floats) *floats = 0;
  if (ints) *ints = 0;
  if (!v2_shape_ok(m, n, k)) return LL_OK;
  const V2Plan pl = v2_plan(n, k);
  if (floats) *floats = (int64_t)pl.nblocks * pl.slots * V2_SLAB;
  if (ints) *ints = (int64_t)pl.nblocks * 4;
  return LL_OK;
}

extern "C" int ll_w4a16_v2_launch(void* out, const void* x, const int32_t* qweight, const float* scales,
                                  const float* zeros, const void* packed, const void* bias, int64_t m, int64_t n, int64_t k,
                                  int group_size, int64_t x_stride_m, int64_t qw_stride_n, int64_t s_stride_n,
                                  float* workspace, int32_t* counters, int epilogue, void* stream) {
  const V2Plan pl = v2_plan(n, k);
  V2Params p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.w = (const uint32_t*)qweight; p.scales = scales;
  p.zeros = zeros; p.packed = packed; p.bias = (const uint16_t*)bias; p.workspace = workspace; p.counters = counters;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m; p.w_stride = qw_stride_n; p.s_stride = s_stride_n;
  p.nblocks = pl.nblocks; p.chunks = pl.chunks; p.total_units = pl.total_units; p.upw = pl.upw; p.slots = pl.slots;
  p.gt = pl.gt; p.gbase = pl.gbase; p.grem = pl.grem; p.glead = pl.glead;
  p.epi = epilogue;
#ifdef V2_TIMELINE
  p.tlwave = getenv("LL_GEMM_TL_WAVE") ? atoi(getenv("LL_GEMM_TL_WAVE")) : 0;
  p.timeline = getenv("LL_GEMM_TIMELINE") ? (unsigned long long*)strtoull(getenv("LL_GEMM_TIMELINE"), nullptr, 16) : nullptr;
#endif
  p.gdiv = group_size / 128;
  p.gshift = -1;
  if ((p.gdiv & (p.gdiv - 1)) == 0) {
    int sh = 0;
    while ((1 << sh) < p.gdiv) ++sh;
    p.gshift = sh;
  }
  hipStream_t st = (hipStream_t)stream;
#define V2_LAUNCH_PK(PK, ABL)                                                                                   \
  {                                                                                                             \
    static bool attr_set_[16] = {false};                                                                        \
    int dev_ = 0;                                                                                               \
    (void)hipGetDevice(&dev_);                                                                                  \
    bool& attr_set = attr_set_[dev_ >= 0 && dev_ < 16 ? dev_ : 0];                                              \
    if (!attr_set) {                                                                                            \
      (void)hipFuncSetAttribute((const void*)wgemm2_kernel<1, PK, ABL>,                                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);                      \
      (void)hipFuncSetAttribute((const void*)wgemm2_kernel<2, PK, ABL>,                                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES);                      \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    if (m <= 32)                                                                                                \
      wgemm2_kernel<1, PK, ABL><<<dim3((unsigned)pl.grid), V2_THREADS, V2_LDS_BYTES, st>>>(p);                  \
    else                                                                                                        \
      wgemm2_kernel<2, PK, ABL><<<dim3((unsigned)pl.grid), V2_THREADS, V2_LDS_BYTES, st>>>(p);                  \
  }
#define V2_LAUNCH(ABL)                                  \
  if (packed) V2_LAUNCH_PK(1, ABL) else V2_LAUNCH_PK(0, ABL)
#ifdef V2_DEBUG_ABLATE
  const int abl = getenv("LL_GEMM2_ABLATE") ? atoi(getenv("LL_GEMM2_ABLATE")) : 0;
  switch (abl) {
    case 1: V2_LAUNCH(1) break;
    case 2: V2_LAUNCH(2) break;
    case 3: V2_LAUNCH(3) break;
    case 4: V2_LAUNCH(4) break;
    case 7: V2_LAUNCH(7) break;
    case 8: V2_LAUNCH(8) break;
    case 12: V2_LAUNCH(12) break;
    case 136: V2_LAUNCH(136) break;
    case 264: V2_LAUNCH(264) break;
    case 139: V2_LAUNCH(139) break;
    case 267: V2_LAUNCH(267) break;
    case 11: V2_LAUNCH(11) break;
    case 13: V2_LAUNCH(13) break;
    case 14: V2_LAUNCH(14) break;
    case 15: V2_LAUNCH(15) break;
    case 23: V2_LAUNCH(23) break;
    case 39: V2_LAUNCH(39) break;
    case 55: V2_LAUNCH(55) break;
    case 71: V2_LAUNCH(71) break;
    default: V2_LAUNCH(0) break;
  }
#else
  V2_LAUNCH(0)
#endif
  return LL_LAUNCH_CHECK();
}
