// Which per-instruction footprint does the CU's vector memory path like?  Every wave sums 256-byte rows of a large
// buffer (row order scrambled), 16 rows per step, with one of three lane -> address maps:
//   mode 0: the MFMA-operand gather of flash_decoding.hip: lane (t = l & 15, c = l >> 4), 4 loads per 16 rows, load s
//           reads bytes [s*64 + c*16, +16) of row t            -> an instruction touches 16 rows x 64 B (half lines)
//   mode 1: full rows: load i reads rows 4i .. 4i+3, lane l -> row 4i + (l >> 4), bytes (l & 15) * 16
//                                                              -> an instruction touches 4 rows x 256 B (whole lines)
//   mode 2: whole 128-byte lines: load (L, g) reads rows 8g .. 8g+7, line L, lane -> row 8g + (t & 7), piece (t >> 3, c)
// hipcc -O3 --offload-arch=gfx950 gather_probe.hip -o gather_probe && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ buf, const int* __restrict__ rows, int steps, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int t = lane & 15, c = lane >> 4;
  const int* my = rows + (size_t)wave * steps * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int st = 0; st < steps; ++st) {
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      size_t off;
      if (MODE == 0) off = (size_t)my[st * 16 + t] * 256 + i * 64 + c * 16;
      else if (MODE == 1) off = (size_t)my[st * 16 + 4 * i + (lane >> 4)] * 256 + (lane & 15) * 16;
      else off = (size_t)my[st * 16 + 8 * (i & 1) + (t & 7)] * 256 + (i >> 1) * 128 + (t >> 3) * 64 + c * 16;
      r[i] = *reinterpret_cast<const u32x4*>(buf + off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += r[i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) out[0] = 1;
}

// mode 3: mode 0 with DEPTH row sets (DEPTH * 4 loads = DEPTH KB) requested before the first is consumed
template <int DEPTH>
__global__ __launch_bounds__(256) void gather_deep(const char* __restrict__ buf, const int* __restrict__ rows, int steps, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int t = lane & 15, c = lane >> 4;
  const int* my = rows + (size_t)wave * steps * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int st = 0; st < steps; st += DEPTH) {
    u32x4 r[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[d][i] = *reinterpret_cast<const u32x4*>(buf + (size_t)my[(st + d) * 16 + t] * 256 + i * 64 + c * 16);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += r[d][i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) out[0] = 1;
}

int main(int argc, char** argv) {
  const size_t nrows = 1u << 22;  // 1 GiB of 256-byte rows
  const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 16, cus = 256, steps = argc > 2 ? atoi(argv[2]) : 64;
  printf("== %d waves per CU, %d steps (x 16 rows x 256 B) per wave\n", waves_per_cu, steps);
  const int waves = waves_per_cu * cus;
  char* buf; int* rows; unsigned* out;
  hipMalloc(&buf, nrows * 256); hipMemset(buf, 1, nrows * 256);
  hipMalloc(&out, 4);
  std::vector<int> h((size_t)waves * steps * 16);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (int)(x % nrows); }
  hipMalloc(&rows, h.size() * 4); hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fill = 0; fill < 2; ++fill) {
  if (fill) {  // second pass: pseudo-random payload instead of a constant byte (does the data pattern matter?)
    std::vector<unsigned> rnd(1u << 24);
    for (auto& v : rnd) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (unsigned)x; }
    for (size_t off = 0; off < nrows * 256; off += rnd.size() * 4) hipMemcpy(buf + off, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice);
    printf("-- random payload --\n");
  }
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) gather<0><<<waves / 4, 256>>>(buf, rows, steps, out);
      if (mode == 1) gather<1><<<waves / 4, 256>>>(buf, rows, steps, out);
      if (mode == 2) gather<2><<<waves / 4, 256>>>(buf, rows, steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)waves * steps * 16 * 256;
      if (rep) printf("mode %d: %.3f ms  %.2f TB/s  (%.1f MB)\n", mode, ms, bytes / ms / 1e9, bytes / 1e6);
    }
  }
  for (int depth = 2; depth <= 8; depth *= 2) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (depth == 2) gather_deep<2><<<waves / 4, 256>>>(buf, rows, steps, out);
      if (depth == 4) gather_deep<4><<<waves / 4, 256>>>(buf, rows, steps, out);
      if (depth == 8) gather_deep<8><<<waves / 4, 256>>>(buf, rows, steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)waves * steps * 16 * 256;
      if (rep) printf("mode 0 x depth %d (%d KB in flight per wave): %.3f ms  %.2f TB/s\n", depth, depth * 4, ms, bytes / ms / 1e9);
    }
  }
  }
  return 0;
}
