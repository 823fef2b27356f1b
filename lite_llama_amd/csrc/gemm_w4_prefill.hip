// W4A16 dequant-GEMM for MANY rows (prefill: M = batch x prompt tokens), over the decode engine's load-time layouts
// (ll_w4a16_pack_weights / ll_w4a16_pack_scales).  Semantics: lite_llama/kernels/quantization/w4a16.py:28-207 -- the reference's
// Triton kernel tiles M x N and dequantises a [BLOCK_N, BLOCK_K] weight tile per k-step for a whole BLOCK_M of rows; the
// decode engines (gemm_w4_v3 / v4) stream the weights once for <= 64 rows and gemm_wq.hip loops that 64-row tile over M
// (32 768 prompt rows re-stream and re-dequantise the weights 512 times: 472 TFLOP/s, round-5 measurement).  Here:
//   * tile 128 (round 6; 256 in round 5: LL_PF_WMH=2) rows of X x 256 weight rows x 64 k; 4 waves = N quarters (8 = 2 M halves x 4), wave tile 128 x 64 -> 8 MFMA
//     32x32x16 accumulators (128 registers); A operand = dequantised weights, B operand = activations (so a lane ends up with
//     consecutive output columns of ONE token row: 16-byte stores after one v_permlane32_swap, the decode engines' epilogue);
//   * the weight tile is dequantised ONCE per 256 token rows: every thread fetches 16 B of the packed stream per k-step
//     (= 32 consecutive k of one weight row, already in MFMA order), runs the exact unpack + fp16 affine map of the decode
//     engines (gemm_w4_common.h: bit-identical dequantisation) and parks 64 B of fp16 in LDS;
//   * the activation tile goes global -> LDS by LDS-DMA (no registers), XOR-swizzled through the per-lane SOURCE address:
//     16-byte slot s of row r sits at s ^ ((r >> 1) & 7) of its 128-byte row -- fragment reads (ds_read_b128, 16-lane groups) and
//     the dequantiser's ds_write_b128 (8-lane groups) are both bank-conflict free;
//   * both tiles double-buffered in LDS (128 KB), one barrier per k-step: the loads of step s + 1 are in flight under the 32
//     MFMAs per wave of step s;
//   * fused epilogues of the decode engines: bias, and silu(gate) * up for the row-interleaved fused gate|up.
#include <stdlib.h>

#include "common.h"
#include "gemm_w4_common.h"

#define PF_BN 256
#define PF_BK 64
#ifndef PF_WMH_DEFAULT
#define PF_WMH_DEFAULT 1  // 128-row M halves per workgroup (see the kernel; measured: 16.3 -> 15.2 ms per Qwen2.5-7B layer at 32 768 rows)
#endif
#ifndef PF_XR
#define PF_XR 3  // activation-tile ring slots (32 KB each): tiles are requested PF_XR - 1 k-steps ahead
#endif

struct PfParams {
  uint16_t* out;
  const uint16_t* x;
  const void* wp;
  const void* sp;
  const uint16_t* bias;
  int64_t m, n, k, x_stride;
  int chunks;   // K / 128
  int gshift;   // log2(group_size / 128)
  int epi;      // 0: out[m, n];  1: rows are (gate_j, up_j) pairs -> out[m, n / 2] = silu(gate) * up
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int pf_swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// WMH: 128-row M halves per workgroup.  2 = the 256 x 256 tile on 8 waves, ONE workgroup per CU (round 5).  1 = a 128 x 256 tile on 4
// waves (round 6): 48 KB of LDS and one wave per SIMD, so two or three workgroups share a CU and one's per-k-step barrier stall is
// covered by the others' MFMAs (the weight fragment is dequantised once per 128 token rows either way).
template <int WMH>
__global__ __launch_bounds__(WMH * 256) __attribute__((amdgpu_waves_per_eu(2, WMH == 2 ? 2 : 3))) void wgemm_prefill_kernel(const PfParams p) {
  constexpr int PF_BM = WMH * 128;
  constexpr int PF_TILE = PF_BM * PF_BK * 2;  // bytes of one [PF_BM][64] fp16 tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const xt = lds;  // [PF_XR][256 token rows][64 k] fp16, swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> tile: groups of 16 x 16 tiles; inside a group workgroup 8 j + xcd (on XCD xcd under the round-robin dispatch)
  // takes an 8 (m) x 4 (n) corner of the group, so one XCD's L2 serves 8 activation tiles and 4 weight tiles per k-step
  int tm, tn;
  {
    const int b = (int)blockIdx.x;
    const int g = b >> 8, i = b & 255, xcd = i & 7, j = i >> 3;
    const int groups_m = (p.tiles_m + 15) >> 4;
    const int gm = g % groups_m, gn = g / groups_m;
    tm = gm * 16 + (xcd & 1) * 8 + (j & 7);
    tn = gn * 16 + (xcd >> 1) * 4 + (j >> 3);
  }
  if (tm >= p.tiles_m || tn >= p.tiles_n) return;
  const int64_t m0 = (int64_t)tm * PF_BM, n0 = (int64_t)tn * PF_BN;
  const int ksteps = (int)(p.k / PF_BK);

  // ---- activation loader role: wave w issues pieces 4w .. 4w + 3 (8 rows x 128 B each); lane -> (row, physical slot) ----
  uint32_t xoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wv * 4 + j) * 8 + (lane >> 3), ps = lane & 7;
    int64_t row = m0 + r;
    if (row >= p.m) row = p.m - 1;  // rows >= M feed only unstored outputs
    xoff[j] = (uint32_t)(row * p.x_stride * 2 + pf_swz(r, ps) * 16);
  }
  auto issue_x = [&](int ks, int buf) {
    const char* xb = (const char*)p.x + (size_t)ks * (PF_BK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) v3_dma16<false>((uint32_t)(buf * PF_TILE + (wv * 4 + j) * 1024), xb, xoff[j]);
  };

  // ---- consumer geometry: wave (wm, wn) multiplies token rows wm * 128 .. + 127 by weight rows wn * 64 .. + 63.  Its weight
  // words come STRAIGHT from the packed stream into registers: lane (nl, h) of piece (128-row tile wn >> 1, row group
  // (wn & 1) * 2 + ni, k-half) holds the four words = k-octets h * 4 + 0..3 of weight row nl -- word kk IS the A operand of MFMA
  // k-step kk (the decode engines' pairing: a k-step contracts octets kk and 4 + kk), so nothing is staged through LDS and
  // nothing is written back; the two waves that share a weight quarter each dequantise it (VALU under the other wave's MFMAs).
  const int wm = WMH == 2 ? wv >> 2 : 0, wn = wv & 3;
  const int nl = lane & 31, h = lane >> 5;
  const char* wsrc[2];
  const char* ssrc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int ng = (wn & 1) * 2 + ni;
    wsrc[ni] = (const char*)p.wp + ((size_t)(n0 / 128 + (wn >> 1)) * p.chunks) * 8192 + (size_t)(ng * 1024 + lane * 16);
    ssrc[ni] = (const char*)p.sp + (size_t)(n0 + wn * 64 + ni * 32 + nl) * 8;
  }
  uint32_t magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  auto load_w = [&](int ks, u32x4 (&w)[2], u32x2 (&sc)[2]) {
    const int c = ks >> 1, kh = ks & 1;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      w[ni] = *reinterpret_cast<const u32x4*>(wsrc[ni] + (size_t)c * 8192 + kh * 4096);
      sc[ni] = *reinterpret_cast<const u32x2*>(ssrc[ni] + (size_t)(c >> p.gshift) * (size_t)p.n * 8);
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.f;
  int xrow_off[4], xsw[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int r = wm * 128 + mi * 32 + nl;
    xrow_off[mi] = r * 128;
    xsw[mi] = (r >> 1) & 7;
  }

  // ---- prologue: the activation tiles of steps 0 .. PF_XR - 2, the weight words of step 0 ----
  u32x4 wA[2], wB[2];
  u32x2 sA[2], sB[2];
  for (int s0 = 0; s0 < PF_XR - 1 && s0 < ksteps; ++s0) issue_x(s0, s0);
  load_w(0, wA, sA);
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): (the builtin -- see the loop)
  __builtin_amdgcn_s_barrier();

#if defined(PF_KEEP) && PF_KEEP  /* A/B: keep the youngest requests (the tile two steps ahead + next step's weight words) in flight */
#define PF_END_WAIT asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
#else
#define PF_END_WAIT __builtin_amdgcn_s_waitcnt(0x0070);
#endif
#define PF_STEP(WC, SC, WN, SN)                                                                                        \
  {                                                                                                                    \
    const unsigned char* xb = xt + (ks % PF_XR) * PF_TILE;                                                             \
    /* fragments + first weight fragments first (the MFMAs can start), then the requests of the steps ahead */        \
    f16x8 xf[2][4], wf[2][2];                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                   \
        xf[0][mi] = *reinterpret_cast<const f16x8*>(xb + xrow_off[mi] + (((h * 4 + 0) ^ xsw[mi]) * 16));               \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) wf[0][ni] = v3_dequant(WC[ni].x, SC[ni].x, SC[ni].y, magic);      \
    if (ks + PF_XR - 1 < ksteps) issue_x(ks + PF_XR - 1, (ks + PF_XR - 1) % PF_XR); /* that buffer was read in step ks - 1 */ \
    if (ks + 1 < ksteps) load_w(ks + 1, WN, SN);                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                                 \
      const int cur = kk & 1;                                                                                          \
      /* k-step kk + 1: activation fragments read and weight fragments dequantised UNDER the eight MFMAs of k-step kk */ \
      if (kk < 3) {                                                                                                    \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                               \
            xf[cur ^ 1][mi] = *reinterpret_cast<const f16x8*>(xb + xrow_off[mi] + (((h * 4 + kk + 1) ^ xsw[mi]) * 16)); \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                             \
          const uint32_t word = kk == 0 ? WC[ni].y : kk == 1 ? WC[ni].z : WC[ni].w;                                    \
          wf[cur ^ 1][ni] = v3_dequant(word, SC[ni].x, SC[ni].y, magic);                                               \
        }                                                                                                              \
      }                                                                                                                \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                               \
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][ni], xf[cur][mi], acc[ni][mi], 0, 0, 0);        \
      if (kk < 3) {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */                                            \
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); /* four VALU of the next dequantisation */                \
          if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* one fragment read */                        \
        }                                                                                                              \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
    PF_END_WAIT                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                      \
    ++ks;                                                                                                              \
  }
  int ks = 0;
  for (;;) {
    PF_STEP(wA, sA, wB, sB)
    if (ks >= ksteps) break;
    PF_STEP(wB, sB, wA, sA)
    if (ks >= ksteps) break;
  }
#undef PF_STEP

  // ---- epilogue (the decode engines'): a lane holds, for token row m = nl, the weight rows 8g + 4h .. + 3 of a 32-row group ----
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
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t ncol = n0 + wn * 64 + ni * 32;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[ni][mi][e];
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
      uint32_t lo[4], hi[4], sw[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f32_to_f16_bits(v[4 * g + e]);
        lo[g] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        hi[g] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        if (p.epi) {  // both GEMM outputs rounded to fp16, then silu(g) * u in fp32: the stand-alone kernels' arithmetic
          const float g0 = f16_bits_to_f32(o[0]), u0 = f16_bits_to_f32(o[1]);
          const float g1 = f16_bits_to_f32(o[2]), u1 = f16_bits_to_f32(o[3]);
          sw[g] = (uint32_t)f32_to_f16_bits(ll_silu_mul_f32(g0, u0)) | ((uint32_t)f32_to_f16_bits(ll_silu_mul_f32(g1, u1)) << 16);
        }
      }
      const int64_t nn = ncol + 16 * h;
      if (p.epi) {
        swap32(sw[0], sw[2]);
        swap32(sw[1], sw[3]);
        if (row_ok) *reinterpret_cast<u32x4*>(p.out + mrow * (p.n >> 1) + (nn >> 1)) = u32x4{sw[0], sw[2], sw[1], sw[3]};
      } else {
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
}

extern "C" int ll_w4a16_mtiled_supported(int64_t m, int64_t n, int64_t k, int group_size) {
  if (m < 1 || n < PF_BN || n % PF_BN != 0 || k < 128 || k % 128 != 0) return 0;
  if (group_size <= 0 || group_size % 128 != 0 || k % group_size != 0) return 0;
  const int gdiv = group_size / 128;
  if (gdiv & (gdiv - 1)) return 0;
  if (n * k / 2 >= (1ll << 31) || n * (k / 128) * 8 >= (1ll << 31)) return 0;  // 32-bit offsets into the packed streams
  const int64_t tiles = ((m + 127) / 128 + 15) / 16 * 16 * ((n / PF_BN + 15) / 16 * 16);
  return tiles < (1ll << 31) ? 1 : 0;
}

// out [m][n] (epilogue 0, + bias) or [m][n / 2] (epilogue 1: fused gate|up rows interleaved) = x [m][k] @ dequant(W)^T over the
// load-time layouts of the decode engine; any m >= 1 (meant for m > 64: the decode engines stream the weights once for fewer rows).
extern "C" int ll_w4a16_matmul_prepacked_mtiled(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias,
                                                int64_t m, int64_t n, int64_t k, int group_size, int64_t x_stride_m, int epilogue,
                                                void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || group_size <= 0) return LL_ERR_SHAPE;
  if (m == 0) return LL_OK;
  if (!ll_w4a16_mtiled_supported(m, n, k, group_size) || x_stride_m % 8 != 0 || (epilogue != 0 && epilogue != 1)) return LL_ERR_SHAPE;
  if (!out || !x || !wpacked || !spacked) return LL_ERR_ARG;
  if (!ll_aligned16(out) || !ll_aligned16(x) || !ll_aligned16(wpacked) || !ll_aligned16(spacked)) return LL_ERR_ARG;
  if ((m - 1) * x_stride_m * 2 + k * 2 >= (1ll << 31)) return LL_ERR_SHAPE;  // 32-bit activation offsets
  PfParams p{};
  p.out = (uint16_t*)out; p.x = (const uint16_t*)x; p.wp = wpacked; p.sp = spacked; p.bias = (const uint16_t*)bias;
  p.m = m; p.n = n; p.k = k; p.x_stride = x_stride_m;
  p.chunks = (int)(k / 128);
  int sh = 0;
  while ((128 << sh) < group_size) ++sh;
  p.gshift = sh;
  p.epi = epilogue;
  static const int wmh_env = getenv("LL_PF_WMH") ? atoi(getenv("LL_PF_WMH")) : 0;  // A/B knob, read once (1 / 2: force the tile height)
  const int wmh = wmh_env == 1 || wmh_env == 2 ? wmh_env : PF_WMH_DEFAULT;
  const int bm = wmh * 128;
  p.tiles_m = (int)((m + bm - 1) / bm);
  p.tiles_n = (int)(n / PF_BN);
  const int64_t grid = (int64_t)((p.tiles_m + 15) / 16) * ((p.tiles_n + 15) / 16) * 256;
  static bool attr_set[16][2] = {{false}};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int lds_bytes = PF_XR * bm * PF_BK * 2;
  if (dev >= 0 && dev < 16 && !attr_set[dev][wmh - 1]) {
    if (wmh == 2) (void)hipFuncSetAttribute((const void*)wgemm_prefill_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    else (void)hipFuncSetAttribute((const void*)wgemm_prefill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set[dev][wmh - 1] = true;
  }
  if (wmh == 2) wgemm_prefill_kernel<2><<<dim3((unsigned)grid), 512, lds_bytes, (hipStream_t)stream>>>(p);
  else wgemm_prefill_kernel<1><<<dim3((unsigned)grid), 256, lds_bytes, (hipStream_t)stream>>>(p);
  return LL_LAUNCH_CHECK();
}
