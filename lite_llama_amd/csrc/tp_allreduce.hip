// One-shot tensor-parallel all-reduce on peer-mapped buffers (the native target for the block's single
// collective -- lite_llama/distributed/parallel_state.py:208-213 issues ncclAllReduce on a [tokens, hidden] fp16 tensor
// after every row-parallel projection, 56 per Qwen2.5-7B step).  The payload is small (0.25-0.5 MB), xGMI is
// point-to-point (every GPU has a direct link to each of its 7 peers): a ring all-reduce pays 2 (N - 1) latency-bound hops,
// a ONE-SHOT exchange pays one -- every rank publishes its partial sums in a buffer its peers have mapped
// (hipIpcMemHandle), raises a flag in each peer's memory, waits for the peers' flags in its OWN memory and then reads the
// N - 1 peer buffers over its N - 1 direct links while adding them up in fp32, in rank order (all ranks produce
// bit-identical sums).  One launch, no host call, capturable in the decode graph.
//
// Protocol per launch (epoch e = *epoch + 1, parity p = e & 1; B = gridDim.x workgroups, workgroup b owns slice b of the
// payload for all three phases, so no grid-wide barrier is needed):
//   1. copy slice b of `inout` into my staging half p (write-through stores), __threadfence_system();
//   2. lane r < world stores e into peer r's flag word [p][b][my rank] (release, system scope);
//   3. lane r < world spins (bounded) on MY flag word [p][b][r] until it reads e (acquire, system scope);
//   4. out[i] = fp16( sum_r fp32(stage_r[p][i]) ) for the slice, peers read with system-scope loads;
//   5. the last workgroup out stores e to *epoch.
// A staging half is rewritten two launches later; a peer can only be one launch behind (it cannot pass the wait of
// launch e + 1 before I have raised flag e + 1, which I do after finishing launch e), so nobody is still reading it.
// Memory: staging and flags live in fine-grained device memory (ll_tp_shared_alloc) so that peer writes become visible
// inside a running kernel.  A wait that times out (~10 s of polling) is FATAL for the job, never silent: the launch sets bit
// 0 of the sticky error word (flags[2 * B * world]) and writes NaN over its slice of the tensor -- the poison reaches the
// logits of this and every later step -- and the callers (DecodeEngine.decode, bench.py, parallel_state.oneshot_error)
// read the word when they synchronise and raise.  (Round 2 summed what had arrived and went on.)
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kMaxWorld = 8;
constexpr int kSpinLimitDefault = 1 << 23;  // ~10 s of polling: ranks may reach their FIRST collective seconds apart (model build)
// LL_TP_SPIN_LOG2 (read once): log2 of the polls a rank waits for a peer's flag.  Only oversubscribed debugging set-ups need
// it (several ranks time-slicing ONE GPU: the spinning ranks burn their polls while the late rank gets 1/N of the device).
static int tp_spin_limit() {
  static const int v = [] {
    const char* e = getenv("LL_TP_SPIN_LOG2");
    const int l = e ? atoi(e) : 0;
    return l >= 10 && l <= 30 ? 1 << l : kSpinLimitDefault;
  }();
  return v;
}

struct PeerTable {
  const uint16_t* stage[kMaxWorld];
  int32_t* flags[kMaxWorld];
};

template <int DT>
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(uint16_t* __restrict__ inout, int64_t count, PeerTable peers,
                                                                int rank, int world, int64_t stage_elems,
                                                                int32_t* __restrict__ epoch, int32_t* __restrict__ done,
                                                                int kSpinLimit) {
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int e = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int p = e & 1;
  const int64_t vecs = (count + 7) / 8;                       // 16-byte pieces
  const int64_t per = (vecs + nb - 1) / nb;
  const int64_t v_lo = (int64_t)b * per, v_hi = v_lo + per < vecs ? v_lo + per : vecs;
  uint16_t* mine = const_cast<uint16_t*>(peers.stage[rank]) + (int64_t)p * stage_elems;
  // 1. publish my slice
  for (int64_t v = v_lo + tid; v < v_hi; v += 256) {
    const i32x4 x = *reinterpret_cast<const i32x4*>(inout + v * 8);
    __builtin_nontemporal_store(x, reinterpret_cast<i32x4*>(mine + v * 8));
  }
  __threadfence_system();
  __syncthreads();
  // 2. + 3. flags
  __shared__ int failed;
  if (tid == 0) failed = 0;
  __syncthreads();
  const int64_t slot = ((int64_t)p * nb + b) * world;
  if (tid < world) {
    __hip_atomic_store(peers.flags[tid] + slot + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    int32_t* my = peers.flags[rank] + slot + tid;
    int spins = 0;
    while (__hip_atomic_load(my, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      if (++spins > kSpinLimit) {
        atomicOr(&failed, 1 << tid);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  const int bad = failed;
  if (bad && tid == 0) atomicOr(peers.flags[rank] + 2 * (int64_t)nb * world, 1);
  // 4. reduce in rank order (a missing peer: poison the slice)
  if (bad) {
    const int nan2 = DT == LL_F16 ? 0x7e007e00 : 0x7fc07fc0;
    for (int64_t v = v_lo + tid; v < v_hi; v += 256) *reinterpret_cast<i32x4*>(inout + v * 8) = i32x4{nan2, nan2, nan2, nan2};
  }
  for (int64_t v = v_lo + tid; v < v_hi && !bad; v += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) {
      const uint16_t* src = peers.stage[r] + (int64_t)p * stage_elems + v * 8;
      const i32x4 x = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(src));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += to_f32<DT>((uint16_t)((uint32_t)x[j] & 0xffffu));
        acc[2 * j + 1] += to_f32<DT>((uint16_t)((uint32_t)x[j] >> 16));
      }
    }
    i32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = (int)((uint32_t)from_f32<DT>(acc[2 * j]) | ((uint32_t)from_f32<DT>(acc[2 * j + 1]) << 16));
    *reinterpret_cast<i32x4*>(inout + v * 8) = o;
  }
  // 5. the last workgroup out advances the epoch
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nb - 1) {
      __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- //
// Row-parallel projection -> all-reduce -> add-and-normalise as ONE launch (round 3).  Under tensor parallelism the
// reference runs GEMM, ncclAllReduce, skip_rmsnorm (lite_llama/models/linear.py:160-161, kernels/skip_rmsnorm.py:192-234);
// at TP = 1 this build already lets the int4 decode GEMM leave fp32 split-K partials [S][rows][n] that the norm adds up
// (gemm_w4_v3.hip epilogue 2, ll_skip_rmsnorm_partials).  Here the same partials feed the collective:
//   A. o_local = fp16(sum_s P[s])      -- exactly what the projection's own epilogue would have stored -- into my staging
//      half (workgroup b owns rows b, b + B, ...: rows, not flat slices, because the norm needs whole rows);
//   B. flags, as in allreduce_oneshot_kernel (same words, same epoch counter: the two kernels may alternate in a step);
//   C. o = fp16(sum_r fp32(stage_r[row]))   in rank order: bit-identical on every rank and to the unfused path;
//   D. skip_rmsnorm: x = o + residual, residual <- x (rounded), y = fp16(x * rsqrt(mean x^2 + eps)) * w.
// One launch instead of GEMM-merge + all-reduce + norm: the TP step keeps the 7 launches per layer of the TP = 1 step.
template <int DT>
__global__ __launch_bounds__(256) void allreduce_norm_partials_kernel(
    uint16_t* __restrict__ y, const float* __restrict__ part, int s_count, uint16_t* __restrict__ resid,
    const uint16_t* __restrict__ w, int64_t rows, int n, float eps, PeerTable peers, int rank, int world,
    int64_t stage_elems, int32_t* __restrict__ epoch, int32_t* __restrict__ done, int kSpinLimit) {
  constexpr int VMAX = 4;  // n <= 8192: up to 4 x 256 pieces of 8 values per row
  __shared__ float red[4];
  __shared__ int failed;
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int e = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int p = e & 1;
  const int nv = n / 8;
  const int64_t plane = rows * (int64_t)n;
  uint16_t* mine = const_cast<uint16_t*>(peers.stage[rank]) + (int64_t)p * stage_elems;
  // A. local sums -> staging
  for (int64_t row = b; row < rows; row += nb) {
    for (int v = tid; v < nv; v += 256) {
      const int64_t off = row * n + (int64_t)v * 8;
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int sl = 0; sl < s_count; ++sl) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(part + sl * plane + off);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(part + sl * plane + off + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] += lo[j];
          a[4 + j] += hi[j];
        }
      }
      i32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (int)((uint32_t)from_f32<DT>(a[2 * j]) | ((uint32_t)from_f32<DT>(a[2 * j + 1]) << 16));
      __builtin_nontemporal_store(o, reinterpret_cast<i32x4*>(mine + off));
    }
  }
  __threadfence_system();
  if (tid == 0) failed = 0;
  __syncthreads();
  // B. flags
  const int64_t slot = ((int64_t)p * nb + b) * world;
  if (tid < world) {
    __hip_atomic_store(peers.flags[tid] + slot + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    int32_t* my = peers.flags[rank] + slot + tid;
    int spins = 0;
    while (__hip_atomic_load(my, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      if (++spins > kSpinLimit) {
        atomicOr(&failed, 1 << tid);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  const int bad = failed;
  if (bad && tid == 0) atomicOr(peers.flags[rank] + 2 * (int64_t)nb * world, 1);
  // C + D per row
  const float nf = (float)n;
  for (int64_t row = b; row < rows; row += nb) {
    float xv[VMAX][8];
    float ssq = 0.f;
#pragma unroll
    for (int it = 0; it < VMAX; ++it) {
      const int v = it * 256 + tid;
      if (v < nv) {
        const int64_t off = row * n + (int64_t)v * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
          const i32x4 x = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(peers.stage[r] + (int64_t)p * stage_elems + off));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[2 * j] += to_f32<DT>((uint16_t)((uint32_t)x[j] & 0xffffu));
            acc[2 * j + 1] += to_f32<DT>((uint16_t)((uint32_t)x[j] >> 16));
          }
        }
        U16x8 rv = *reinterpret_cast<const U16x8*>(resid + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = to_f32<DT>(from_f32<DT>(acc[j])) + to_f32<DT>(rv.v[j]);
          rv.v[j] = bad ? (uint16_t)(DT == LL_F16 ? 0x7e00 : 0x7fc0) : from_f32<DT>(x);
          xv[it][j] = x;
          ssq += x * x / nf;
        }
        *reinterpret_cast<U16x8*>(resid + off) = rv;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[it][j] = 0.f;
      }
    }
    // block sum of ssq (4 waves)
    ssq = wave_sum(ssq);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    const float var = red[0] + red[1] + red[2] + red[3];
    const float rrms = bad ? __builtin_nanf("") : 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int it = 0; it < VMAX; ++it) {
      const int v = it * 256 + tid;
      if (v < nv) {
        const U16x8 wv = *reinterpret_cast<const U16x8*>(w + (int64_t)v * 8);
        U16x8 yv;
#pragma unroll
        for (int j = 0; j < 8; ++j) yv.v[j] = mul_storage<DT>(from_f32<DT>(xv[it][j] * rrms), wv.v[j]);
        *reinterpret_cast<U16x8*>(y + row * n + (int64_t)v * 8) = yv;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nb - 1) {
      __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

// Fine-grained device memory (peer stores become visible inside a running kernel), zero-filled.
extern "C" int ll_tp_shared_alloc(void** ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return LL_ERR_ARG;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    return LL_ERR_LAUNCH;
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return LL_ERR_LAUNCH;
  }
  *ptr = p;
  return LL_OK;
}
extern "C" int ll_tp_shared_free(void* ptr) { return ptr && hipFree(ptr) != hipSuccess ? LL_ERR_LAUNCH : LL_OK; }

// 64-byte handle of an allocation for another process / its mapping here (peer access enabled lazily).
extern "C" int ll_tp_ipc_export(void* ptr, void* handle64) {
  if (!ptr || !handle64) return LL_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  return hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr) == hipSuccess ? LL_OK : LL_ERR_LAUNCH;
}
extern "C" int ll_tp_ipc_open(const void* handle64, void** ptr) {
  if (!ptr || !handle64) return LL_ERR_ARG;
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle64, sizeof(h));
  return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? LL_OK : LL_ERR_LAUNCH;
}
extern "C" int ll_tp_ipc_close(void* ptr) { return ptr && hipIpcCloseMemHandle(ptr) != hipSuccess ? LL_ERR_LAUNCH : LL_OK; }

// The error word of a flag allocation (its LAST word), read back by the host; synchronises the device.
extern "C" int ll_tp_error_word(const void* flags, int64_t flag_words, int32_t* host_out) {
  if (!flags || !host_out || flag_words < 1) return LL_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return LL_ERR_LAUNCH;
  return hipMemcpy(host_out, (const int32_t*)flags + flag_words - 1, sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess
             ? LL_OK : LL_ERR_LAUNCH;
}

// Flag words a rank needs for `blocks` workgroups: [2][blocks][world] + the error word.
extern "C" int64_t ll_tp_oneshot_flag_words(int blocks, int world) { return 2ll * blocks * world + 1; }

// In-place SUM over the ranks of a TP group.  stage_ptrs / flag_ptrs: HOST arrays of `world` device pointers (entry r =
// rank r's staging buffer [2][stage_elems] 16-bit / flag words, mine included; peers' entries come from ll_tp_ipc_open);
// epoch_done: two int32 in MY memory (launch counter, workgroup ticket), zero before the first launch; every rank must
// issue the same sequence of calls with the same count and blocks.  count <= stage_elems, count % 8 == 0.
extern "C" int ll_tp_allreduce_oneshot(void* inout, int64_t count, int dtype, const void* const* stage_ptrs,
                                       void* const* flag_ptrs, int rank, int world, int64_t stage_elems, int blocks,
                                       int32_t* epoch_done, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || count < 0 || count > stage_elems || count % 8 != 0 ||
      blocks < 1 || blocks > 1024)
    return LL_ERR_SHAPE;
  if (count == 0) return LL_OK;
  if (!inout || !stage_ptrs || !flag_ptrs || !epoch_done || !ll_aligned16(inout)) return LL_ERR_ARG;
  PeerTable t{};
  for (int r = 0; r < world; ++r) {
    if (!stage_ptrs[r] || !flag_ptrs[r]) return LL_ERR_ARG;
    t.stage[r] = (const uint16_t*)stage_ptrs[r];
    t.flags[r] = (int32_t*)flag_ptrs[r];
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16)
    allreduce_oneshot_kernel<LL_F16><<<blocks, 256, 0, st>>>((uint16_t*)inout, count, t, rank, world, stage_elems,
                                                              epoch_done, epoch_done + 1, tp_spin_limit());
  else
    allreduce_oneshot_kernel<LL_BF16><<<blocks, 256, 0, st>>>((uint16_t*)inout, count, t, rank, world, stage_elems,
                                                               epoch_done, epoch_done + 1, tp_spin_limit());
  return LL_LAUNCH_CHECK();
}

// y = skip_rmsnorm(allreduce_tp(fp16(sum of the S fp32 partials [S][rows][n] of a row-parallel projection)), residual, weight):
// the projection's split-K partials, the one-shot all-reduce and the add-and-normalise in one launch (residual updated in
// place, as ll_skip_rmsnorm does).  Same staging / flag / epoch words and the same `blocks` as ll_tp_allreduce_oneshot (the
// two may alternate); rows * n <= stage_elems, n % 8 == 0, n <= 8192, 1 <= s_count <= 64.
extern "C" int ll_tp_allreduce_norm_partials(void* y, const float* partials, int s_count, void* residual, const void* weight,
                                             int64_t rows, int64_t n, float eps, int dtype, const void* const* stage_ptrs,
                                             void* const* flag_ptrs, int rank, int world, int64_t stage_elems, int blocks,
                                             int32_t* epoch_done, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || rows < 0 || n <= 0 || n % 8 != 0 || n > 8192 ||
      rows * n > stage_elems || s_count < 1 || s_count > 64 || blocks < 1 || blocks > 1024)
    return LL_ERR_SHAPE;
  if (rows == 0) return LL_OK;
  if (!y || !partials || !residual || !weight || !stage_ptrs || !flag_ptrs || !epoch_done || !ll_aligned16(y) ||
      !ll_aligned16(partials) || !ll_aligned16(residual) || !ll_aligned16(weight))
    return LL_ERR_ARG;
  PeerTable t{};
  for (int r = 0; r < world; ++r) {
    if (!stage_ptrs[r] || !flag_ptrs[r]) return LL_ERR_ARG;
    t.stage[r] = (const uint16_t*)stage_ptrs[r];
    t.flags[r] = (int32_t*)flag_ptrs[r];
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LL_F16)
    allreduce_norm_partials_kernel<LL_F16><<<blocks, 256, 0, st>>>((uint16_t*)y, partials, s_count, (uint16_t*)residual,
                                                                    (const uint16_t*)weight, rows, (int)n, eps, t, rank, world,
                                                                    stage_elems, epoch_done, epoch_done + 1, tp_spin_limit());
  else
    allreduce_norm_partials_kernel<LL_BF16><<<blocks, 256, 0, st>>>((uint16_t*)y, partials, s_count, (uint16_t*)residual,
                                                                     (const uint16_t*)weight, rows, (int)n, eps, t, rank, world,
                                                                     stage_elems, epoch_done, epoch_done + 1, tp_spin_limit());
  return LL_LAUNCH_CHECK();
}
