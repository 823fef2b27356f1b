// Load-time ingestion of third-party int4 checkpoint layouts into the path's native W4A16 layout
// (qweight [N, K/8] int32, nibbles sequential along K, LSB first; scales / zeros fp32 [N, K/g] --
// reference lite_llama/kernels/quantization/w4a16.py:152-207).  The reference cannot load these
// checkpoints at all (models/weights.py:166-173,266-268); the formats restated here
// are the published ones of
//   * AutoAWQ 0.2.x "GEMM" (WQLinear_GEMM): qweight int32 [K, N/8], eight OUTPUT channels per word
//     in the order {0, 2, 4, 6, 1, 3, 5, 7} (nibble i of word j holds channel 8j + order[i]);
//     qzeros int32 [K/g, N/8] packed the same way; scales fp16 [K/g, N]; w = (q - z) * s
//   * AutoGPTQ 0.7.x (QuantLinear, v1 checkpoints): qweight int32 [K/8, N], eight INPUT channels per
//     word, sequential, LSB first; qzeros int32 [K/g, N/8] sequential holding z - 1; scales fp16
//     [K/g, N]; g_idx[k] = k / g (no activation reordering); w = (q - (zs + 1)) * s
// Pure integer / byte moves + one exact fp16 -> fp32 widening: bit-exact by construction.
// HBM-bound shuffles; tiles go through LDS so that both the reads and the writes are coalesced.
#include "common.h"

namespace {

// nibble position inside an AWQ word that holds output channel (n % 8)
__device__ __forceinline__ int awq_shift(int c) {
  // inverse of {0,2,4,6,1,3,5,7}: channel c sits at nibble (c >> 1) + 4 * (c & 1)
  return 4 * ((c >> 1) + 4 * (c & 1));
}

// ---- qweight: AWQ [K, N/8] -> native [N, K/8] ------------------------------------------------
// Block = 32 column-words (256 output channels) x 8 k-octets (64 k).  Phase 1: thread (ng, ko) reads
// its 8 words (k = 8 ko + i, ng) -- consecutive threads read consecutive words of a row -- and
// transposes the 8x8 nibble square in registers into 8 native words (n = 8 ng + c, ko), parked in
// LDS as tile[n_local][ko].  Phase 2: rows of the tile go out with ko fastest (32 B per row here;
// the block's 8 k-octets are contiguous in the native row).
__global__ __launch_bounds__(256) void awq_qweight_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                          int64_t k, int64_t n) {
  __shared__ uint32_t tile[256][9];
  const int64_t nw = n / 8, kw = k / 8;
  const int ng_l = threadIdx.x & 31, ko_l = threadIdx.x >> 5;
  const int64_t ng = (int64_t)blockIdx.x * 32 + ng_l, ko = (int64_t)blockIdx.y * 8 + ko_l;
  if (ng < nw && ko < kw) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = in[(ko * 8 + i) * nw + ng];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int sh = 4 * ((c >> 1) + 4 * (c & 1));
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) o |= ((w[i] >> sh) & 0xFu) << (4 * i);
      tile[ng_l * 8 + c][ko_l] = o;
    }
  }
  __syncthreads();
  const int ko2 = threadIdx.x & 7;
  const int64_t kq = (int64_t)blockIdx.y * 8 + ko2;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int nl = (threadIdx.x >> 3) + 32 * r;
    const int64_t nn = (int64_t)blockIdx.x * 256 + nl;
    if (nn < n && kq < kw) out[nn * kw + kq] = tile[nl][ko2];
  }
}

// ---- qweight: GPTQ [K/8, N] -> native [N, K/8]: a 32-bit matrix transpose -----------------------
__global__ __launch_bounds__(256) void gptq_qweight_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                           int64_t kw, int64_t n) {
  __shared__ uint32_t tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t n0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t kk = k0 + ty + 8 * r, nn = n0 + tx;
    if (kk < kw && nn < n) tile[ty + 8 * r][tx] = in[kk * n + nn];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t nn = n0 + ty + 8 * r, kk = k0 + tx;
    if (kk < kw && nn < n) out[nn * kw + kk] = tile[tx][ty + 8 * r];
  }
}

// ---- scales fp16 [G, N] -> fp32 [N, G]; zeros (packed int32 [G, N/8]) -> fp32 [N, G] -------------
// MODE 0: AWQ nibble order, z as stored; MODE 1: sequential order, z + zero_offset.
template <int MODE>
__global__ __launch_bounds__(256) void w4_scales_zeros_kernel(float* __restrict__ out_s, float* __restrict__ out_z,
                                                              const uint16_t* __restrict__ scales,
                                                              const uint32_t* __restrict__ qzeros, int64_t groups,
                                                              int64_t n, int zero_offset) {
  __shared__ float ts[32][33];
  __shared__ float tz[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t n0 = (int64_t)blockIdx.x * 32, g0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t g = g0 + ty + 8 * r, nn = n0 + tx;
    if (g < groups && nn < n) {
      ts[ty + 8 * r][tx] = f16_bits_to_f32(scales[g * n + nn]);
      const uint32_t w = qzeros[g * (n / 8) + nn / 8];
      const int c = (int)(nn & 7);
      const int sh = MODE == 0 ? awq_shift(c) : 4 * c;
      tz[ty + 8 * r][tx] = (float)(int)(((w >> sh) & 0xFu) + (uint32_t)zero_offset);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t nn = n0 + ty + 8 * r, g = g0 + tx;
    if (g < groups && nn < n) {
      out_s[nn * groups + g] = ts[tx][ty + 8 * r];
      out_z[nn * groups + g] = tz[tx][ty + 8 * r];
    }
  }
}

int check_common(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f, int64_t k,
                 int64_t n, int64_t group) {
  if (!a || !b || !c || !d || !e || !f) return LL_ERR_ARG;
  if (k <= 0 || n <= 0 || group <= 0 || k % 8 != 0 || n % 8 != 0 || k % group != 0) return LL_ERR_SHAPE;
  return LL_OK;
}

}  // namespace

extern "C" int ll_w4_from_awq(int32_t* out_qweight, float* out_scales, float* out_zeros, const int32_t* qweight,
                              const int32_t* qzeros, const void* scales_f16, int64_t k, int64_t n, int64_t group_size,
                              void* stream) {
  const int rc = check_common(out_qweight, out_scales, out_zeros, qweight, qzeros, scales_f16, k, n, group_size);
  if (rc != LL_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nw = n / 8, kw = k / 8, groups = k / group_size;
  awq_qweight_kernel<<<dim3((unsigned)((nw + 31) / 32), (unsigned)((kw + 7) / 8)), 256, 0, st>>>(
      (uint32_t*)out_qweight, (const uint32_t*)qweight, k, n);
  w4_scales_zeros_kernel<0><<<dim3((unsigned)((n + 31) / 32), (unsigned)((groups + 31) / 32)), 256, 0, st>>>(
      out_scales, out_zeros, (const uint16_t*)scales_f16, (const uint32_t*)qzeros, groups, n, 0);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_w4_from_gptq(int32_t* out_qweight, float* out_scales, float* out_zeros, const int32_t* qweight,
                               const int32_t* qzeros, const void* scales_f16, int64_t k, int64_t n, int64_t group_size,
                               int zero_offset, void* stream) {
  const int rc = check_common(out_qweight, out_scales, out_zeros, qweight, qzeros, scales_f16, k, n, group_size);
  if (rc != LL_OK) return rc;
  if (zero_offset < 0 || zero_offset > 1) return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t kw = k / 8, groups = k / group_size;
  gptq_qweight_kernel<<<dim3((unsigned)((n + 31) / 32), (unsigned)((kw + 31) / 32)), 256, 0, st>>>(
      (uint32_t*)out_qweight, (const uint32_t*)qweight, kw, n);
  w4_scales_zeros_kernel<1><<<dim3((unsigned)((n + 31) / 32), (unsigned)((groups + 31) / 32)), 256, 0, st>>>(
      out_scales, out_zeros, (const uint16_t*)scales_f16, (const uint32_t*)qzeros, groups, n, zero_offset);
  return LL_LAUNCH_CHECK();
}
