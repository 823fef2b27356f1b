// Probe (round 6): can a DEPENDENT kernel be launched early on a forked graph branch, request its weights, and wait for its
// predecessor through a device flag -- does hipGraph co-run the two branches, what does the pair cost against the serial form?
// K1 = a streaming kernel (256 workgroups x 320 threads, reads `k1_mb` MB, writes a small vector X write-through, arrives on a counter).
// K2 = a short weight-stream consumer (256 x 768 threads, 112 KB LDS): requests 28 KB of weights per workgroup into registers,
// [waits for the counter], reads X with coherent loads, writes a result.  28 (K1, K2) pairs per graph; K1(i+1) depends on K2(i).
// Bounded spins: a wait that times out (1 ms) raises a flag instead of hanging.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/elp benchmarks/probes/early_launch_probe.hip ; run: /tmp/elp [k1_mb]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(320) void k1(const u32x4* __restrict__ src, size_t n16_per_wg, float* x, int xn, unsigned* counter, float salt) {
  const u32x4* p = src + (size_t)blockIdx.x * n16_per_wg;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < n16_per_wg; i += 320 * 4) {
    u32x4 a = __builtin_nontemporal_load(p + i);
    u32x4 b = i + 320 < n16_per_wg ? __builtin_nontemporal_load(p + i + 320) : a;
    u32x4 c = i + 640 < n16_per_wg ? __builtin_nontemporal_load(p + i + 640) : a;
    u32x4 d = i + 960 < n16_per_wg ? __builtin_nontemporal_load(p + i + 960) : a;
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  // output: xn floats per workgroup, written through (sc1)
  for (int t = threadIdx.x; t < xn; t += 320) {
    float v = salt + (float)blockIdx.x + (float)t * 0.001f + (acc == 0x12345u ? 1.f : 0.f);
    if (counter) __hip_atomic_store(x + (size_t)blockIdx.x * xn + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else x[(size_t)blockIdx.x * xn + t] = v;
  }
  if (counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool WAIT>
__global__ __launch_bounds__(768) void k2(const u32x4* __restrict__ w, const float* x, int xn, float* out, unsigned* counter, unsigned* done,
                                          unsigned* err, unsigned target) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x;
  // "weights": 2 x 16 B per thread of the 8 consumer waves = 16 KB ... x 2 rounds  (28 KB / workgroup like the o projection)
  u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
  if (tid < 512) {
    const u32x4* p = w + (size_t)blockIdx.x * 1536 + tid;
    a = __builtin_nontemporal_load(p);
    b = __builtin_nontemporal_load(p + 512);
    c = __builtin_nontemporal_load(p + 1024);
  }
  if (WAIT) {
    if (tid == 512) {  // lane 0 of loader wave 0
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 100000ull) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
  }
  // X: every workgroup reads all 256 * xn floats (coherent loads when waiting) -- 458 KB from L2 / MALL
  float s = 0.f;
#pragma unroll 8
  for (int i = tid; i < 256 * xn; i += 768) {
    float v;
    if (WAIT) v = __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v = x[i];
    s += v;
  }
  float* red = (float*)lds;
  red[tid] = s + (float)((a.x ^ b.x ^ c.x) == 0x7654321u);
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
    for (int i = tid; i < 768; i += 64) t += red[i];
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
    red[tid] = t;
  }
  if (tid == 0) {
    float t = red[0];
    out[blockIdx.x] = t;
    if (WAIT) {
      const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {  // every workgroup of this launch is past its wait: re-arm for the next pair
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

int main(int argc, char** argv) {
  const double k1_mb = argc > 1 ? atof(argv[1]) : 68.8;
  const int xn = 448, pairs = 28, WG = 256;  // X = 256 * 448 floats = 458 KB (the attention output's size)
  const size_t n16 = (size_t)(k1_mb * 1e6 / 16 / WG);
  u32x4 *src, *w;
  float *x, *out;
  unsigned* ctr;
  CK(hipMalloc(&src, n16 * WG * 16 * 2));
  CK(hipMalloc(&w, (size_t)WG * 1536 * 16 * pairs));
  CK(hipMalloc(&x, (size_t)WG * xn * 4));
  CK(hipMalloc(&out, WG * 4 * pairs));
  CK(hipMalloc(&ctr, 64 * 4));
  CK(hipMemset(src, 1, n16 * WG * 16 * 2));
  CK(hipMemset(w, 2, (size_t)WG * 1536 * 16 * pairs));
  CK(hipMemset(ctr, 0, 64 * 4));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(4 * pairs);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipFuncSetAttribute((const void*)k2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 114688);
  hipFuncSetAttribute((const void*)k2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 114688);

  auto expected = [&](int pair) {  // sum over all X of the pair
    double t = 0;
    for (int b = 0; b < WG; ++b)
      for (int i = 0; i < xn; ++i) t += (float)((float)pair + (float)b + (float)i * 0.001f);
    return t;
  };
  for (int mode = 0; mode < 3; ++mode) {  // 0: serial, 1: forked + wait (K2 captured first), 2: forked + wait (K1 captured first)
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < pairs; ++p) {
      const u32x4* sp = src + (p & 1) * n16 * WG;
      const u32x4* wp = w + (size_t)p * WG * 1536;
      if (mode == 0) {
        k1<<<WG, 320, 0, s0>>>(sp, n16, x, xn, nullptr, (float)p);
        k2<false><<<WG, 768, 114688, s0>>>(wp, x, xn, out + p * WG, ctr, ctr + 16, ctr + 32, WG);
      } else {
        CK(hipEventRecord(ev[4 * p], s0));
        CK(hipStreamWaitEvent(s1, ev[4 * p], 0));
        if (mode == 1) {
          k2<true><<<WG, 768, 114688, s1>>>(wp, x, xn, out + p * WG, ctr, ctr + 16, ctr + 32, WG);
          k1<<<WG, 320, 0, s0>>>(sp, n16, x, xn, ctr, (float)p);
        } else {
          k1<<<WG, 320, 0, s0>>>(sp, n16, x, xn, ctr, (float)p);
          k2<true><<<WG, 768, 114688, s1>>>(wp, x, xn, out + p * WG, ctr, ctr + 16, ctr + 32, WG);
        }
        CK(hipEventRecord(ev[4 * p + 1], s1));
        CK(hipStreamWaitEvent(s0, ev[4 * p + 1], 0));
      }
    }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    const int reps = 20;
    CK(hipEventRecord(t0, s0));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(t1, s0));
    CK(hipStreamSynchronize(s0));
    float ms;
    CK(hipEventElapsedTime(&ms, t0, t1));
    std::vector<float> h(WG * pairs);
    unsigned hc[64];
    CK(hipMemcpy(h.data(), out, WG * 4 * pairs, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, ctr, sizeof hc, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int p = 0; p < pairs; ++p) {
      const double e = expected(p);
      for (int b = 0; b < WG; ++b) if (fabs(h[p * WG + b] - e) > 1e-4 * e) ++bad;
    }
    printf("mode %d (%s): %.2f us per (K1, K2) pair, %.1f us per graph; wrong outputs %d / %d; timeout flag %u, counter %u done %u\n", mode,
           mode == 0 ? "serial, no flag" : mode == 1 ? "forked, K2 captured first" : "forked, K1 captured first",
           ms * 1e3 / reps / pairs, ms * 1e3 / reps, bad, WG * pairs, hc[32], hc[0], hc[16]);
    CK(hipMemset(ctr, 0, 64 * 4));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  // K1 alone / K2 alone (serial chains) for reference
  for (int which = 0; which < 2; ++which) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < pairs; ++p) {
      if (which == 0) k1<<<WG, 320, 0, s0>>>(src + (p & 1) * n16 * WG, n16, x, xn, nullptr, (float)p);
      else k2<false><<<WG, 768, 114688, s0>>>(w + (size_t)p * WG * 1536, x, xn, out + p * WG, ctr, ctr + 16, ctr + 32, WG);
    }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(t0, s0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(t1, s0));
    CK(hipStreamSynchronize(s0));
    float ms;
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("%s alone: %.2f us per launch\n", which == 0 ? "K1" : "K2", ms * 1e3 / 20 / pairs);
  }
  return 0;
}
