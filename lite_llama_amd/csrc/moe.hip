// fused_moe pieces (a11) -- reference lite_llama/kernels/fused_moe.py:
//   moe_align_block_size :45-99   (torch ops in the reference; one sync-free HIP kernel here)
//   _fused_moe_kernel    :105-211 (grouped GEMM over expert-sorted, block-padded slots)
// The vLLM data protocol is kept bit-exactly: sorted_token_ids (slot ids stably sorted by
// expert, every expert's run padded to block_size with the sentinel num_slots), expert_ids
// (expert per row block; searchsorted(right=True) clamped to E-1 for the unused tail) and
// num_tokens_post_padded (device scalar; row blocks past it exit early -> no host sync).
//
// Grouped GEMM on MFMA 32x32x16: expert weights (fp16, or fp8-e4m3 / int8 with one scale per
// group_n x group_k block) are the streamed "A" operand in fragment layout exactly as in
// gemm_wq.hip; the gathered activation rows of one row block are staged through LDS.
#include <stdlib.h>

#include "common.h"

struct alignas(16) Q4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ int64_t moe_load_idx(const void* p, int64_t i, int w) {
  return w == LL_I32 ? (int64_t)((const int32_t*)p)[i] : ((const int64_t*)p)[i];
}

// ---------------------------------------------------------------------------------- //
// moe_align_block_size: one workgroup, no host sync.  Thread e owns expert e (loops when
// E > 1024): count, padded prefix, then a stable in-order placement scan.
// ---------------------------------------------------------------------------------- //
// An id outside [0, experts) -- never produced by a router over finite logits; the reference indexes out of bounds with it
// (fused_moe.py:214-233) -- is folded onto the nearest valid expert instead of indexing LDS / sorted_ids out of bounds
// (a NaN-poisoned activation reaches the router before the host raises).
__device__ __forceinline__ int moe_clamp_id(int e, int experts) { return e < 0 ? 0 : (e >= experts ? experts - 1 : e); }

__global__ __launch_bounds__(1024) void moe_align_kernel(const void* __restrict__ topk_ids, int ids_w,
                                                         int num_slots, int num_experts, int block_size,
                                                         int32_t* __restrict__ sorted_ids,
                                                         int32_t* __restrict__ expert_ids,
                                                         int32_t* __restrict__ num_post, int max_padded,
                                                         int max_blocks, int stage_ids) {
  extern __shared__ int sm[];  // counts[E], starts[E], block_ends[E]
  int* counts = sm;
  int* starts = sm + num_experts;
  int* bends = sm + 2 * num_experts;
  int* ids_lds = stage_ids ? sm + 3 * num_experts : nullptr;
  const int tid = threadIdx.x;
  for (int i = tid; i < max_padded; i += 1024) sorted_ids[i] = num_slots;  // sentinel
  for (int e = tid; e < num_experts; e += 1024) counts[e] = 0;
  __syncthreads();
  for (int i = tid; i < num_slots; i += 1024) {
    const int e = moe_clamp_id((int)moe_load_idx(topk_ids, i, ids_w), num_experts);
    if (ids_lds) ids_lds[i] = e;
    atomicAdd(&counts[e], 1);
  }
  __syncthreads();
  // padded prefix over the experts: a workgroup scan per 1024 experts (thread e owns expert base + e), the running total
  // carried between rounds -- not one thread walking all experts (that walk was 9 of the kernel's 28 us at E = 128)
  {
    __shared__ int wave_tot[16], carry_s;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < num_experts; base += 1024) {
      const int e = base + tid;
      const int padded = e < num_experts ? (counts[e] + block_size - 1) / block_size * block_size : 0;
      int x = padded;
      const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
      }
      if (lane == 63) wave_tot[wave] = x;
      __syncthreads();
      int before = carry_s;
      for (int w = 0; w < wave; ++w) before += wave_tot[w];
      if (e < num_experts) {
        starts[e] = before + x - padded;
        bends[e] = (before + x) / block_size;
      }
      __syncthreads();
      if (tid == 1023) carry_s = before + x;
      __syncthreads();
    }
    if (tid == 0) num_post[0] = carry_s;
  }
  __syncthreads();
  // stable placement (token order kept inside each expert)
  if (ids_lds) {
    // decode-sized batches: the ids sit in LDS and every SLOT finds its rank among the earlier slots of
    // the same expert (all lanes read the same word per step: an LDS broadcast) -- num_slots short
    // iterations instead of one thread per expert walking global memory (57 us -> a few us at 512 slots)
    // (eight independent LDS reads per round: a one-read-per-iteration loop pays the LDS latency 511 times for the last
    // slot -- 15 of the kernel's 19 us; reads past slot i are masked, the staging area is padded to a multiple of 8)
    for (int i = tid; i < num_slots; i += 1024) {
      const int e = ids_lds[i];
      int rank = 0;
      for (int j0 = 0; j0 < i; j0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ids_lds[j0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (j0 + u < i) & (v[u] == e);
      }
      sorted_ids[starts[e] + rank] = i;
    }
  } else {
    // any size: expert e walks the slots in order
    for (int e = tid; e < num_experts; e += 1024) {
      if (counts[e] == 0) continue;
      int dst = starts[e];
      for (int i = 0; i < num_slots; ++i)
        if (moe_clamp_id((int)moe_load_idx(topk_ids, i, ids_w), num_experts) == e) sorted_ids[dst++] = i;
    }
  }
  // expert_ids[b] = searchsorted(block_ends, b, right=True) clamped to E-1
  for (int b = tid; b < max_blocks; b += 1024) {
    int lo = 0, hi = num_experts;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bends[mid] <= b) lo = mid + 1; else hi = mid;
    }
    expert_ids[b] = lo < num_experts - 1 ? lo : num_experts - 1;
  }
}

// Decode-sized batches (<= 1024 slots, <= 1024 experts): one thread per SLOT.  Round 6: every expert keeps a BITMAP of the slots
// that chose it (one 32-bit word per half-wave of slots, filled with one LDS atomic OR per thread -- order-independent); the stable
// rank of a slot among the earlier slots of its expert is the number of bits below it, an expert's total the number of bits.
// Four barriers, no serial loop over slots or distinct experts (round 5 ranked with one wave-ballot round per DISTINCT expert of
// a wave -- ~50 rounds at top-8 of 128 experts: 8 us per call, 48 calls per Qwen3-30B-A3B step; the first form scanned
// quadratically: 12 us).  Same outputs, bit for bit.
// COHERENT: the ids were written through by OTHER workgroups of this launch (the fused router, below): int64, read with sc1 loads
template <bool COHERENT>
__device__ __forceinline__ void moe_align_small_body(const void* __restrict__ topk_ids, int ids_w, int num_slots, int num_experts,
                                                     int block_size, int32_t* __restrict__ sorted_ids, int32_t* __restrict__ expert_ids,
                                                     int32_t* __restrict__ num_post, int max_padded, int max_blocks, int* sm) {
  // sm: bm[E][2 waves] (32-bit words: which slots of half-wave j chose expert e) | starts[E] | bends[E]
  // (32-bit on purpose: the dynamic LDS region starts wherever the kernel's static objects end -- a 64-bit DS access there faults)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nw = nthreads >> 6, nh = 2 * nw;
  unsigned* bm = reinterpret_cast<unsigned*>(sm);
  int* starts = sm + nh * num_experts;
  int* bends = starts + num_experts;
  __shared__ int wave_tot[16];
  for (int i = tid; i < max_padded; i += nthreads) sorted_ids[i] = num_slots;  // sentinel
  for (int i = tid; i < nh * num_experts; i += nthreads) bm[i] = 0u;
  __syncthreads();
  const bool has = tid < num_slots;
  const int half = tid >> 5;  // this slot's half-wave = its bitmap word
  int e = -1;
  if (has) {
    if constexpr (COHERENT)
      e = moe_clamp_id((int)__hip_atomic_load((const int64_t*)topk_ids + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), num_experts);
    else
      e = moe_clamp_id((int)moe_load_idx(topk_ids, tid, ids_w), num_experts);
    atomicOr(bm + e * nh + half, 1u << (tid & 31));  // (an OR: the result does not depend on the order of arrival)
  }
  __syncthreads();
  // the stable rank of a slot among its expert's slots = bits below it in the expert's bitmap; thread t owns expert t: its total,
  // then the padded prefix over the experts
  int rank = 0;
  if (has) {
    const unsigned* row = bm + e * nh;
    for (int j = 0; j < half; ++j) rank += __popc(row[j]);
    rank += __popc(row[half] & ((1u << (tid & 31)) - 1u));
  }
  const int t = tid;
  int count = 0;
  if (t < num_experts)
    for (int j = 0; j < nh; ++j) count += __popc(bm[t * nh + j]);
  const int padded = t < num_experts ? (count + block_size - 1) / block_size * block_size : 0;
  int x = padded;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_tot[wave] = x;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
  if (t < num_experts) {
    starts[t] = before + x - padded;
    bends[t] = (before + x) / block_size;
  }
  if (tid == nthreads - 1) num_post[0] = before + x;
  __syncthreads();
  if (has) sorted_ids[starts[e] + rank] = tid;
  for (int b = tid; b < max_blocks; b += nthreads) {  // expert_ids[b] = searchsorted(block_ends, b, right=True) clamped to E-1
    int lo = 0, hi = num_experts;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bends[mid] <= b) lo = mid + 1; else hi = mid;
    }
    expert_ids[b] = lo < num_experts - 1 ? lo : num_experts - 1;
  }
}

static size_t moe_align_small_lds(int threads, int num_experts) {
  return (size_t)(threads / 32) * num_experts * sizeof(unsigned) + (size_t)2 * num_experts * sizeof(int);
}

__global__ __launch_bounds__(1024) void moe_align_small_kernel(const void* __restrict__ topk_ids, int ids_w, int num_slots,
                                                               int num_experts, int block_size, int32_t* __restrict__ sorted_ids,
                                                               int32_t* __restrict__ expert_ids, int32_t* __restrict__ num_post,
                                                               int max_padded, int max_blocks) {
  extern __shared__ int sm[];
  moe_align_small_body<false>(topk_ids, ids_w, num_slots, num_experts, block_size, sorted_ids, expert_ids, num_post, max_padded,
                              max_blocks, sm);
}

// ---------------------------------------------------------------------------------- //
// Prefill-sized inputs (round 6: thousands to hundreds of thousands of slots).  The one-workgroup kernel above walks ALL slots
// once per expert thread: 21.9 ms per call at 32 768 tokens x top-8 of 128 experts -- 60 % of the Qwen3-30B-A3B prefill pass.
// Three launches over a caller-provided workspace instead, same outputs bit for bit:
//   count    one workgroup per chunk of 1024 slots: the bitmap ranking of the small kernel inside the chunk (one LDS atomic OR per
//            slot), per-(chunk, expert) counts -> workspace[chunk][expert]; sentinel fill of sorted_ids by all workgroups;
//   offsets  one workgroup: exclusive prefix of the counts over the chunks (per expert, in place), the padded prefix over the
//            experts (-> workspace[chunks * E + e]), expert_ids, num_tokens_post_padded;
//   place    one workgroup per chunk: bitmaps again (the ids are 1 - 2 MB), sorted_ids[start[e] + before[chunk][e] + rank] = slot.
// Stable by construction: chunks are consecutive slot ranges, ranks inside a chunk count the earlier slots of the same expert.
// ---------------------------------------------------------------------------------- //
#define MOE_ALIGN_CHUNK 1024
// bitmaps of one chunk: bm[E][32 half-waves]; returns this thread's expert (-1: no slot) and its rank among the chunk's earlier
// slots of that expert
__device__ __forceinline__ int moe_align_chunk_bitmaps(const void* __restrict__ topk_ids, int ids_w, int num_slots, int num_experts,
                                                       unsigned* bm, int& rank) {
  const int tid = threadIdx.x, slot = (int)blockIdx.x * MOE_ALIGN_CHUNK + tid;
  constexpr int NH = MOE_ALIGN_CHUNK / 32;
  for (int i = tid; i < NH * num_experts; i += MOE_ALIGN_CHUNK) bm[i] = 0u;
  __syncthreads();
  const int half = tid >> 5;
  int e = -1;
  if (slot < num_slots) {
    e = moe_clamp_id((int)moe_load_idx(topk_ids, slot, ids_w), num_experts);
    atomicOr(bm + e * NH + half, 1u << (tid & 31));
  }
  __syncthreads();
  rank = 0;
  if (e >= 0) {
    const unsigned* row = bm + e * NH;
    for (int j = 0; j < half; ++j) rank += __popc(row[j]);
    rank += __popc(row[half] & ((1u << (tid & 31)) - 1u));
  }
  return e;
}

__global__ __launch_bounds__(MOE_ALIGN_CHUNK) void moe_align_count_kernel(const void* __restrict__ topk_ids, int ids_w, int num_slots,
                                                                          int num_experts, int32_t* __restrict__ sorted_ids,
                                                                          int max_padded, int32_t* __restrict__ ws) {
  extern __shared__ int sm[];
  unsigned* bm = reinterpret_cast<unsigned*>(sm);
  constexpr int NH = MOE_ALIGN_CHUNK / 32;
  for (int i = (int)blockIdx.x * MOE_ALIGN_CHUNK + threadIdx.x; i < max_padded; i += (int)gridDim.x * MOE_ALIGN_CHUNK)
    sorted_ids[i] = num_slots;  // sentinel
  int rank;
  (void)moe_align_chunk_bitmaps(topk_ids, ids_w, num_slots, num_experts, bm, rank);
  for (int e = threadIdx.x; e < num_experts; e += MOE_ALIGN_CHUNK) {
    int c = 0;
    for (int j = 0; j < NH; ++j) c += __popc(bm[e * NH + j]);
    ws[(size_t)blockIdx.x * num_experts + e] = c;
  }
}

__global__ __launch_bounds__(1024) void moe_align_offsets_kernel(int num_experts, int block_size, int chunks, int32_t* __restrict__ ws,
                                                                 int32_t* __restrict__ expert_ids, int32_t* __restrict__ num_post,
                                                                 int max_blocks) {
  extern __shared__ int sm[];  // part_tot[P][E] | bends[E]
  const int tid = threadIdx.x;
  // thread (part, e): chunks [part * per, ...) of expert e -- P = 1024 / E parts walk the chunk axis side by side
  const int P = num_experts <= 1024 ? 1024 / num_experts : 1;
  const int per = (chunks + P - 1) / P;
  int* part_tot = sm;
  int* bends = sm + P * num_experts;
  __shared__ int wave_tot[16], carry_s;
  if (tid == 0) carry_s = 0;
  for (int base = 0; base < num_experts; base += 1024) {  // (one round unless E > 1024: then P = 1)
    const int part = num_experts <= 1024 ? tid / num_experts : 0;
    const int e = num_experts <= 1024 ? tid - part * num_experts : base + tid;
    const bool live = e < num_experts && part < P;
    const int c_lo = part * per, c_hi = c_lo + per < chunks ? c_lo + per : chunks;
    int tot = 0;
    if (live) {
      int c = c_lo;
      for (; c + 8 <= c_hi; c += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(c + u) * num_experts + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += v[u];
      }
      for (; c < c_hi; ++c) tot += ws[(size_t)c * num_experts + e];
      part_tot[part * num_experts + e] = tot;
    }
    __syncthreads();
    int before_parts = 0, count = 0;
    if (live) {
      for (int q = 0; q < P; ++q) {
        const int t = part_tot[q * num_experts + e];
        if (q < part) before_parts += t;
        count += t;
      }
      // exclusive prefix over this part's chunks, in place
      int run = before_parts, c = c_lo;
      for (; c + 8 <= c_hi; c += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(c + u) * num_experts + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ws[(size_t)(c + u) * num_experts + e] = run;
          run += v[u];
        }
      }
      for (; c < c_hi; ++c) {
        const int v = ws[(size_t)c * num_experts + e];
        ws[(size_t)c * num_experts + e] = run;
        run += v;
      }
    }
    // padded prefix over the experts (threads of part 0 carry the counts; every other thread contributes 0)
    const int padded = (live && part == 0) ? (count + block_size - 1) / block_size * block_size : 0;
    int x = padded;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (live && part == 0) {
      ws[(size_t)chunks * num_experts + e] = before + x - padded;  // start of expert e's run
      bends[e] = (before + x) / block_size;
    }
    __syncthreads();
    if (tid == 1023) carry_s = before + x;
    __syncthreads();
  }
  if (tid == 0) num_post[0] = carry_s;
  for (int b = tid; b < max_blocks; b += 1024) {  // expert_ids[b] = searchsorted(block_ends, b, right=True) clamped to E-1
    int lo = 0, hi = num_experts;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bends[mid] <= b) lo = mid + 1; else hi = mid;
    }
    expert_ids[b] = lo < num_experts - 1 ? lo : num_experts - 1;
  }
}

__global__ __launch_bounds__(MOE_ALIGN_CHUNK) void moe_align_place_kernel(const void* __restrict__ topk_ids, int ids_w, int num_slots,
                                                                          int num_experts, int chunks, int32_t* __restrict__ sorted_ids,
                                                                          const int32_t* __restrict__ ws) {
  extern __shared__ int sm[];
  int rank;
  const int e = moe_align_chunk_bitmaps(topk_ids, ids_w, num_slots, num_experts, reinterpret_cast<unsigned*>(sm), rank);
  if (e >= 0)
    sorted_ids[ws[(size_t)chunks * num_experts + e] + ws[(size_t)blockIdx.x * num_experts + e] + rank] =
        (int)blockIdx.x * MOE_ALIGN_CHUNK + threadIdx.x;
}

// int32 words of workspace the multi-workgroup form wants for this input; 0: the one-workgroup kernels serve it (decode-sized
// inputs, or more experts than a chunk's bitmaps fit in 64 KB of LDS for)
extern "C" int64_t ll_moe_align_workspace_ints(int64_t num_slots, int num_experts) {
  static const bool off = getenv("LL_MOE_ALIGN_ONE_WG") != nullptr;  // A/B knob, read once
  if (off || num_slots <= 1024 || num_experts <= 0 || (size_t)num_experts * (MOE_ALIGN_CHUNK / 32) * 4 > 64 * 1024) return 0;
  if (num_experts > 1024 && (size_t)num_experts * 8 > 64 * 1024) return 0;
  const int64_t chunks = (num_slots + MOE_ALIGN_CHUNK - 1) / MOE_ALIGN_CHUNK;
  return chunks * num_experts + num_experts;
}

extern "C" int ll_moe_align_block_size(const void* topk_ids, int ids_width, int64_t num_slots, int num_experts, int block_size,
                                       int32_t* sorted_ids, int32_t* expert_ids, int32_t* num_post, void* stream);

extern "C" int ll_moe_align_block_size_ws(const void* topk_ids, int ids_width, int64_t num_slots, int num_experts, int block_size,
                                          int32_t* sorted_ids, int32_t* expert_ids, int32_t* num_post, int32_t* workspace,
                                          int64_t workspace_ints, void* stream) {
  const int64_t want = ll_moe_align_workspace_ints(num_slots, num_experts);
  if (want == 0 || !workspace || workspace_ints < want || (ids_width != LL_I32 && ids_width != LL_I64) ||
      block_size <= 0 || num_slots + (int64_t)num_experts * (block_size - 1) >= (1ll << 31))
    return ll_moe_align_block_size(topk_ids, ids_width, num_slots, num_experts, block_size, sorted_ids, expert_ids, num_post, stream);
  const int max_padded = (int)num_slots + num_experts * (block_size - 1);
  const int max_blocks = (max_padded + block_size - 1) / block_size;
  const int chunks = (int)((num_slots + MOE_ALIGN_CHUNK - 1) / MOE_ALIGN_CHUNK);
  const size_t bm_bytes = (size_t)num_experts * (MOE_ALIGN_CHUNK / 32) * sizeof(unsigned);
  const int P = num_experts <= 1024 ? 1024 / num_experts : 1;
  hipStream_t st = (hipStream_t)stream;
  moe_align_count_kernel<<<chunks, MOE_ALIGN_CHUNK, bm_bytes, st>>>(topk_ids, ids_width, (int)num_slots, num_experts, sorted_ids,
                                                                    max_padded, workspace);
  moe_align_offsets_kernel<<<1, 1024, (size_t)(P + 1) * num_experts * sizeof(int), st>>>(num_experts, block_size, chunks, workspace,
                                                                                         expert_ids, num_post, max_blocks);
  moe_align_place_kernel<<<chunks, MOE_ALIGN_CHUNK, bm_bytes, st>>>(topk_ids, ids_width, (int)num_slots, num_experts, chunks,
                                                                    sorted_ids, workspace);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_moe_align_block_size(const void* topk_ids, int ids_width, int64_t num_slots, int num_experts,
                                       int block_size, int32_t* sorted_ids, int32_t* expert_ids,
                                       int32_t* num_post, void* stream) {
  if (ids_width != LL_I32 && ids_width != LL_I64) return LL_ERR_DTYPE;
  if (num_slots < 0 || num_experts <= 0 || block_size <= 0 || num_experts > 8192) return LL_ERR_SHAPE;
  const int max_padded = (int)num_slots + num_experts * (block_size - 1);
  const int max_blocks = (max_padded + block_size - 1) / block_size;
  static const bool small_off = getenv("LL_MOE_ALIGN_V1") != nullptr;  // A/B knob, read once
  const int small_threads = (int)((num_slots > num_experts ? num_slots : num_experts) + 63) / 64 * 64;
  // (per-(expert, wave) bitmaps in dynamic LDS: kept inside the 64 KB a launch gets without an opt-in attribute; larger shapes
  // take the general kernel)
  const size_t small_lds = moe_align_small_lds(small_threads, num_experts);
  if (!small_off && num_slots >= 1 && num_slots <= 1024 && num_experts <= 1024 && small_lds <= 64 * 1024) {
    int threads = small_threads;
    moe_align_small_kernel<<<1, threads, small_lds, (hipStream_t)stream>>>(
        topk_ids, ids_width, (int)num_slots, num_experts, block_size, sorted_ids, expert_ids, num_post, max_padded, max_blocks);
    return LL_LAUNCH_CHECK();
  }
  const int stage_ids = num_slots <= 4096 ? 1 : 0;  // the rank scan is quadratic: decode-sized batches only
  moe_align_kernel<<<1, 1024, (3 * num_experts + (stage_ids ? (int)num_slots + 8 : 0)) * sizeof(int), (hipStream_t)stream>>>(
      topk_ids, ids_width, (int)num_slots, num_experts, block_size, sorted_ids, expert_ids, num_post, max_padded,
      max_blocks, stage_ids);
  return LL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------- //
// grouped GEMM
// ---------------------------------------------------------------------------------- //
__device__ __forceinline__ uint32_t mpk_mul(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t mpk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t mpk_bcast(float v) {
  const uint32_t h = f32_to_f16_bits(v);
  return h | (h << 16);
}
__device__ __forceinline__ void mdq_i8(uint32_t w, uint32_t s, uint32_t& o0, uint32_t& o1) {
  const uint32_t u = w ^ 0x80808080u;
  o0 = mpk_mul(mpk_add(__builtin_amdgcn_perm(0x64646464u, u, 0x04010400u), 0xE480E480u), s);
  o1 = mpk_mul(mpk_add(__builtin_amdgcn_perm(0x64646464u, u, 0x04030402u), 0xE480E480u), s);
}
__device__ __forceinline__ void mdq_fp8(uint32_t w, uint32_t s256, uint32_t& o0, uint32_t& o1) {
  uint32_t p0 = __builtin_amdgcn_perm(0u, w, 0x010C000Cu);
  uint32_t p1 = __builtin_amdgcn_perm(0u, w, 0x030C020Cu);
  p0 = (p0 & 0x80008000u) | ((p0 >> 1) & 0x3F803F80u);  // fused_moe.py:191-201 (same bit trick)
  p1 = (p1 & 0x80008000u) | ((p1 >> 1) & 0x3F803F80u);
  o0 = mpk_mul(p0, s256);
  o1 = mpk_mul(p1, s256);
}

struct MoeParams {
  uint16_t* c;            // [num_slots, N]
  const uint16_t* a;      // [tokens or slots, K]
  const void* w;          // [E, N, K] fp16 / uint8 / int8
  const float* w_scale;   // [E, ceil(N/gn), ceil(K/gk)] or null
  const uint16_t* topk_w; // [num_slots] in the activation dtype
  const int32_t* sorted_ids;
  const int32_t* expert_ids;
  const int32_t* num_post;
  int64_t num_slots, n, k;
  int block_m, top_k, mul_w, group_n;
  int64_t group_k;
  int64_t a_stride, w_stride_e, w_stride_n, s_stride_e, s_stride_n, s_stride_k;
};

// grid = (ceil(N/128), EM / block_m), block = 256.  WFMT: LL_W_F16 / LL_W_FP8E4M3 / LL_W_INT8.
// ADT: activation dtype (LL_F16 only on the MFMA f16 path; bf16 activations use the bf16 MFMA).
template <int WFMT, int MT>
__global__ __launch_bounds__(256) void moe_gemm_kernel(const MoeParams p) {
  constexpr int A_ROW_BYTES = 256 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[MT * 32 * A_ROW_BYTES];
  __shared__ int32_t row_slot[MT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nl = lane & 31, h = lane >> 5;
  const int pid_m = blockIdx.y;
  if ((int64_t)pid_m * p.block_m >= p.num_post[0]) return;  // fused_moe.py:160-162
  const int expert = p.expert_ids[pid_m];
  if (tid < MT * 32) {
    const int32_t s = tid < p.block_m ? p.sorted_ids[(int64_t)pid_m * p.block_m + tid] : (int32_t)p.num_slots;
    row_slot[tid] = s;
  }
  __syncthreads();

  const int64_t n0 = (int64_t)blockIdx.x * 128 + wv * 32;
  int64_t nrow = n0 + nl;
  if (nrow >= p.n) nrow = p.n - 1;
  const int K = (int)p.k;
  constexpr int EB = (WFMT == LL_W_F16) ? 2 : 1;  // bytes per weight element
  const unsigned char* wrow = (const unsigned char*)p.w + ((int64_t)expert * p.w_stride_e + nrow * p.w_stride_n) * EB;
  const float* srow = p.w_scale ? p.w_scale + (int64_t)expert * p.s_stride_e + (nrow / p.group_n) * p.s_stride_n : nullptr;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // activation staging: thread (row = tid/8, seg = tid%8): 16 k = 32 B ... 32*MT rows x 128 k
  const int chunks = (K + 127) / 128;
  for (int c = 0; c < chunks; ++c) {
    __syncthreads();
    for (int i = tid; i < MT * 32 * 16; i += 256) {
      const int row = i >> 4, q = i & 15;
      const int kk = c * 128 + q * 8;
      const int32_t slot = row_slot[row];
      Q4 v = Q4{0, 0, 0, 0};
      if (slot < p.num_slots && kk < K)
        v = *reinterpret_cast<const Q4*>(p.a + (int64_t)(slot / p.top_k) * p.a_stride + kk);
      *reinterpret_cast<Q4*>(lds + row * A_ROW_BYTES + q * 16) = v;
    }
    __syncthreads();
    const int kbase = c * 128 + h * 64;
    uint32_t sp = 0;
    // scale blocks along k: one per 64-k half when group_k is a multiple of 64 (or covers K); finer grids -- a
    // tensor-parallel shard that cuts the checkpoint's 128-blocks and re-expresses them at gcd(128, shard) -- per 8-k step
    const bool fine_k = p.group_k % 64 != 0 && p.group_k < K;
    if constexpr (WFMT != LL_W_F16) {
      const int kq = kbase < K ? kbase : K - 1;
      const float sv = srow[(int64_t)(kq / p.group_k) * p.s_stride_k];
      sp = mpk_bcast(WFMT == LL_W_FP8E4M3 ? sv * 256.0f : sv);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = kbase + s * 8;
      Q4 wf = Q4{0, 0, 0, 0};
      if constexpr (WFMT != LL_W_F16) {
        if (fine_k) {
          const int kq = kk < K ? kk : K - 1;
          const float sv = srow[(int64_t)(kq / p.group_k) * p.s_stride_k];
          sp = mpk_bcast(WFMT == LL_W_FP8E4M3 ? sv * 256.0f : sv);
        }
      }
      if (kk < K) {
        if constexpr (WFMT == LL_W_F16) {
          wf = *reinterpret_cast<const Q4*>(wrow + (int64_t)kk * 2);
        } else {
          const uint2 raw = *reinterpret_cast<const uint2*>(wrow + kk);
          if constexpr (WFMT == LL_W_FP8E4M3) {
            mdq_fp8(raw.x, sp, wf.x, wf.y);
            mdq_fp8(raw.y, sp, wf.z, wf.w);
          } else {
            mdq_i8(raw.x, sp, wf.x, wf.y);
            mdq_i8(raw.y, sp, wf.z, wf.w);
          }
        }
      }
      const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f16x8 afrag = *reinterpret_cast<const f16x8*>(lds + (mt * 32 + nl) * A_ROW_BYTES + h * 128 + s * 16);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, afrag, acc[mt], 0, 0, 0);
      }
    }
  }

  // epilogue: D[n][m]: lane = row-block column m (nl), rows n = 8g + 4h + e
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + nl;
    const int32_t slot = row_slot[m];
    if (m >= p.block_m || slot >= p.num_slots) continue;
    float rw = 1.f;
    if (p.mul_w & 1) rw = f16_bits_to_f32(p.topk_w[slot]);  // router weight multiplies in fp32 (:203-205)
    if (p.mul_w & 2) {  // rows are (gate_j, up_j) pairs: silu(g) * u in the epilogue (see moe_gemm_kernel2)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t nn = (int64_t)blockIdx.x * 128 + wv * 32 + 8 * g + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          if (nn + e + 1 < p.n) {
            const float gv = f16_bits_to_f32(f32_to_f16_bits(acc[mt][4 * g + e] * rw));
            const float uv = f16_bits_to_f32(f32_to_f16_bits(acc[mt][4 * g + e + 1] * rw));
            p.c[(int64_t)slot * (p.n >> 1) + ((nn + e) >> 1)] = f32_to_f16_bits(ll_silu_mul_f32(gv, uv));
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t nn = (int64_t)blockIdx.x * 128 + wv * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nn + e < p.n) p.c[(int64_t)slot * p.n + nn + e] = f32_to_f16_bits(acc[mt][4 * g + e] * rw);
    }
  }
}

// Second form (K % 128 == 0, 16-byte aligned weight rows): the 128 x 128 weight tile of a chunk is fetched with FULL-LINE
// loads -- 8 lanes cover one row's 128 (256) contiguous bytes, the next chunk's loads are in flight under the current
// chunk's MFMAs -- and handed to the MFMA layout through LDS, instead of every lane picking 8-byte pieces out of its own
// row at a 2-KB stride (64 cache lines per load instruction, 2.3 TB/s on the Qwen3-30B-A3B expert stack).  The expert
// stack is read exactly once per step whatever the routing, so this is the kernel's whole cost.  Same arithmetic, same
// epilogue as the form above.
#ifndef MOE_NT_W
#define MOE_NT_W 1
#endif
template <int WFMT, int MT>
__global__ __launch_bounds__(256) void moe_gemm_kernel2(const MoeParams p) {
  constexpr int EB = (WFMT == LL_W_F16) ? 2 : 1;        // bytes per weight element
  constexpr int A_ROW_BYTES = 256 + 16;
  constexpr int W_ROW_BYTES = 128 * EB + 16;
  constexpr int WP = 8 * EB;                             // 16-byte pieces per weight row and chunk
  constexpr int WPASS = 128 * WP / 256;                  // 4 (8-bit) or 8 (fp16) pieces per thread
  constexpr int APASS = MT * 32 * 16 / 256;              // activation pieces per thread
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[MT * 32 * A_ROW_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char lds_w[128 * W_ROW_BYTES];
  __shared__ int32_t row_slot[MT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nl = lane & 31, h = lane >> 5;
  const int pid_m = blockIdx.y;
  if ((int64_t)pid_m * p.block_m >= p.num_post[0]) return;
  const int expert = p.expert_ids[pid_m];
  if (tid < MT * 32) row_slot[tid] = tid < p.block_m ? p.sorted_ids[(int64_t)pid_m * p.block_m + tid] : (int32_t)p.num_slots;
  __syncthreads();

  const int64_t ntile = (int64_t)blockIdx.x * 128;
  const int K = (int)p.k;
  const unsigned char* wbase = (const unsigned char*)p.w + (int64_t)expert * p.w_stride_e * EB;
  // loader role: piece (row, q) of the tile, pass by pass
  const unsigned char* wsrc[WPASS];
  int wdst[WPASS];
#pragma unroll
  for (int ps = 0; ps < WPASS; ++ps) {
    const int i = ps * 256 + tid, row = i / WP, q = i % WP;
    int64_t nrow = ntile + row;
    if (nrow >= p.n) nrow = p.n - 1;                     // rows past N feed outputs that are never stored
    wsrc[ps] = wbase + nrow * p.w_stride_n * EB + q * 16;
    wdst[ps] = row * W_ROW_BYTES + q * 16;
  }
  const uint16_t* asrc[APASS];
  int adst[APASS];
  bool aok[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int i = ps * 256 + tid, row = i >> 4, q = i & 15;
    const int32_t slot = row_slot[row];
    aok[ps] = slot < p.num_slots;
    asrc[ps] = p.a + (int64_t)(aok[ps] ? slot / p.top_k : 0) * p.a_stride + q * 8;
    adst[ps] = row * A_ROW_BYTES + q * 16;
  }
  // consumer role: weight row n0 + nl of wave wv's 32-row group
  int64_t crow = ntile + wv * 32 + nl;
  if (crow >= p.n) crow = p.n - 1;
  const float* srow = p.w_scale ? p.w_scale + (int64_t)expert * p.s_stride_e + (crow / p.group_n) * p.s_stride_n : nullptr;
  const unsigned char* wfrag_base = lds_w + (wv * 32 + nl) * W_ROW_BYTES + h * (64 * EB);
  const unsigned char* afrag_base = lds_a + nl * A_ROW_BYTES + h * 128;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  i32x4 wreg[WPASS], areg[APASS];
  auto fetch = [&](int c) {
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
      // the expert stack is read exactly once per step: non-temporal (MOE_NT_W; round 4, as the dense engines' big streams)
      if constexpr (MOE_NT_W != 0) wreg[ps] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wsrc[ps] + (int64_t)c * (128 * EB)));
      else wreg[ps] = *reinterpret_cast<const i32x4*>(wsrc[ps] + (int64_t)c * (128 * EB));
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps)  // rows without a token all re-read ONE piece (no traffic): their LDS rows are zeroed
      areg[ps] = *reinterpret_cast<const i32x4*>(aok[ps] ? asrc[ps] + c * 128 : p.a);
  };
  const int chunks = K / 128;
  fetch(0);
  for (int c = 0; c < chunks; ++c) {
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) *reinterpret_cast<i32x4*>(lds_w + wdst[ps]) = wreg[ps];
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) *reinterpret_cast<i32x4*>(lds_a + adst[ps]) = aok[ps] ? areg[ps] : i32x4{0, 0, 0, 0};
    __syncthreads();
    fetch(c + 1 < chunks ? c + 1 : c);  // in flight under this chunk's arithmetic (unconditional: no branch around loads)
    const int kbase = c * 128 + h * 64;
    uint32_t sp = 0;
    if constexpr (WFMT != LL_W_F16) {
      const float sv = srow[(int64_t)(kbase / p.group_k) * p.s_stride_k];
      sp = mpk_bcast(WFMT == LL_W_FP8E4M3 ? sv * 256.0f : sv);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      Q4 wf;
      if constexpr (WFMT == LL_W_F16) {
        wf = *reinterpret_cast<const Q4*>(wfrag_base + s * 16);
      } else {
        const uint2 raw = *reinterpret_cast<const uint2*>(wfrag_base + s * 8);
        if constexpr (WFMT == LL_W_FP8E4M3) {
          mdq_fp8(raw.x, sp, wf.x, wf.y);
          mdq_fp8(raw.y, sp, wf.z, wf.w);
        } else {
          mdq_i8(raw.x, sp, wf.x, wf.y);
          mdq_i8(raw.y, sp, wf.z, wf.w);
        }
      }
      const f16x8 wfrag = __builtin_bit_cast(f16x8, wf);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f16x8 afrag = *reinterpret_cast<const f16x8*>(afrag_base + mt * 32 * A_ROW_BYTES + s * 16);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag, afrag, acc[mt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + nl;
    const int32_t slot = row_slot[m];
    if (m >= p.block_m || slot >= p.num_slots) continue;
    float rw = 1.f;
    if (p.mul_w & 1) rw = f16_bits_to_f32(p.topk_w[slot]);
    if (p.mul_w & 2) {
      // rows are (gate_j, up_j) pairs (load-time interleave of gate|up, model.py::SparseMoeBlock): both GEMM outputs rounded to
      // fp16, then silu(g) * u -- the arithmetic of silu_and_mul on the [slots, 2 I] tensor the unfused route stores
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t nn = ntile + wv * 32 + 8 * g + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          if (nn + e + 1 < p.n) {
            const float gv = f16_bits_to_f32(f32_to_f16_bits(acc[mt][4 * g + e] * rw));
            const float uv = f16_bits_to_f32(f32_to_f16_bits(acc[mt][4 * g + e + 1] * rw));
            p.c[(int64_t)slot * (p.n >> 1) + ((nn + e) >> 1)] = f32_to_f16_bits(ll_silu_mul_f32(gv, uv));
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t nn = ntile + wv * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nn + e < p.n) p.c[(int64_t)slot * p.n + nn + e] = f32_to_f16_bits(acc[mt][4 * g + e] * rw);
    }
  }
}

extern "C" int ll_moe_gemm(void* c, const void* a, const void* w, const float* w_scale, const void* topk_w,
                           const int32_t* sorted_ids, const int32_t* expert_ids, const int32_t* num_post,
                           int64_t num_slots, int64_t em, int block_m, int64_t n, int64_t k, int top_k,
                           int mul_routed_weight, int wfmt, int group_n, int64_t group_k, int64_t a_stride_m,
                           int64_t w_stride_e, int64_t w_stride_n, int64_t s_stride_e, int64_t s_stride_n,
                           int64_t s_stride_k, int dtype, void* stream) {
  if (dtype != LL_F16) return LL_ERR_DTYPE;  // the reference MoE path is fp16 end to end
  if (wfmt != LL_W_F16 && wfmt != LL_W_FP8E4M3 && wfmt != LL_W_INT8) return LL_ERR_DTYPE;
  if (block_m != 16 && block_m != 32 && block_m != 64 && block_m != 128) return LL_ERR_SHAPE;
  if (n <= 0 || k <= 0 || top_k <= 0 || k % 8 != 0 || a_stride_m % 8 != 0) return LL_ERR_SHAPE;
  if (mul_routed_weight < 0 || mul_routed_weight > 3 || ((mul_routed_weight & 2) && (n & 1))) return LL_ERR_SHAPE;
  if (wfmt != LL_W_F16 && (!w_scale || group_n <= 0 || group_k <= 0)) return LL_ERR_ARG;
  if (wfmt != LL_W_F16 && group_k < k && group_k % 8 != 0) return LL_ERR_SHAPE;  // a scale block spans whole 8-k MFMA steps
  if (wfmt == LL_W_F16 ? (w_stride_n % 8 != 0) : (w_stride_n % 8 != 0)) return LL_ERR_SHAPE;
  if (num_slots == 0) return LL_OK;
  MoeParams p{};
  p.c = (uint16_t*)c; p.a = (const uint16_t*)a; p.w = w; p.w_scale = w_scale; p.topk_w = (const uint16_t*)topk_w;
  p.sorted_ids = sorted_ids; p.expert_ids = expert_ids; p.num_post = num_post;
  p.num_slots = num_slots; p.n = n; p.k = k; p.block_m = block_m; p.top_k = top_k; p.mul_w = mul_routed_weight;
  p.group_n = group_n > 0 ? group_n : 1; p.group_k = group_k > 0 ? group_k : 1;
  p.a_stride = a_stride_m; p.w_stride_e = w_stride_e; p.w_stride_n = w_stride_n;
  p.s_stride_e = s_stride_e; p.s_stride_n = s_stride_n; p.s_stride_k = s_stride_k;
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)((em + block_m - 1) / block_m));
  hipStream_t st = (hipStream_t)stream;
  static const bool v1_forced = getenv("LL_MOE_V1") != nullptr;  // A/B knob, read once
  const bool lines = !v1_forced && k % 128 == 0 && (w_stride_n * (wfmt == LL_W_F16 ? 2 : 1)) % 16 == 0 &&
                     (w_stride_e * (wfmt == LL_W_F16 ? 2 : 1)) % 16 == 0 && ll_aligned16(w) && ll_aligned16(a) &&
                     (wfmt == LL_W_F16 || group_k % 64 == 0);
  // 128-row blocks (round 6: prefill-sized inputs, hundreds of rows per expert): the full-line form only -- a weight tile is
  // dequantised once for four MFMA row tiles and re-read from L2 half as often as with 64-row blocks
  if (block_m == 128 && !lines) return LL_ERR_SHAPE;
#define LL_MOE(WF)                                                                  \
  if (lines) {                                                                      \
    if (block_m == 128) moe_gemm_kernel2<WF, 4><<<grid, 256, 0, st>>>(p);           \
    else if (block_m == 64) moe_gemm_kernel2<WF, 2><<<grid, 256, 0, st>>>(p);       \
    else moe_gemm_kernel2<WF, 1><<<grid, 256, 0, st>>>(p);                          \
  } else if (block_m == 64) moe_gemm_kernel<WF, 2><<<grid, 256, 0, st>>>(p);        \
  else moe_gemm_kernel<WF, 1><<<grid, 256, 0, st>>>(p)
  if (wfmt == LL_W_F16) { LL_MOE(LL_W_F16); }
  else if (wfmt == LL_W_FP8E4M3) { LL_MOE(LL_W_FP8E4M3); }
  else { LL_MOE(LL_W_INT8); }
#undef LL_MOE
  return LL_LAUNCH_CHECK();
}


// ---------------------------------------------------------------------------------- //
// Router tail in one launch (round 3): lite_llama/models/qwen3_moe.py:85-100 runs, after the fp16 router GEMM,
// softmax(dtype=fp32) over ALL experts -> top-k -> renormalise -> cast to the activation dtype as four tensor ops per
// layer (x 48 layers in Qwen3-30B-A3B).  One wave per token: the row's logits live in registers (experts / 64 per lane),
// max and sum are wave reductions, the k selections are k rounds of a wave arg-max over (probability bits, ~index) --
// probabilities are positive floats, so their bit patterns order like the values, and equal values resolve to the LOWER
// expert index (torch.topk leaves the order among equal values unspecified; this kernel is deterministic).
// ---------------------------------------------------------------------------------- //

// (Round 4 also built the WHOLE router -- gate GEMM + this tail -- as one launch, twice: every lane streaming half a gate row
// with v_dot2 (64 cache lines per load instruction), then MFMA with the contraction split over the four waves of a 16-token
// workgroup and wave-private LDS pipelines.  22 us per call against 9.4 us for the library GEMM + the tail below at batch 64
// x 128 experts x 2048: a workgroup cannot pull the 0.5 MB gate matrix through one CU faster than ~12 us, and the four
// sequential tails per wave cost another 5.  benchmarks/probes/moe_router_one_launch.patch.)
// Wave reductions on DPP / permlane moves (a handful of cycles per step) instead of __shfl_xor (ds_bpermute: ~100 cycles each,
// six dependent ones per reduction): the k selection rounds of the router tail are nothing but reductions -- with shuffles a
// token's tail took ~5 us (eight rounds of twelve dependent shuffles on a 64-bit key).
template <int CTRL>
__device__ __forceinline__ uint32_t rt_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// OP over the 16 lanes of a row (quad swaps, then rotations by 4 and 8), then across the four rows
// (every moved value lands in a temporary BEFORE it is combined, and the combiners are branch-free: a DPP move under a
// partial EXEC mask -- what a `a > b ? a : b` on the moved value compiles to -- reads disabled lanes)
#define RT_WAVE_REDUCE(NAME, T, OP, TO_U, FROM_U)                                                              \
  __device__ __forceinline__ T NAME(T x) {                                                                     \
    T y;                                                                                                       \
    y = FROM_U(rt_dpp<0xB1>(TO_U(x)));  x = OP(x, y);  /* quad_perm [1,0,3,2] */                                \
    y = FROM_U(rt_dpp<0x4E>(TO_U(x)));  x = OP(x, y);  /* quad_perm [2,3,0,1] */                                \
    y = FROM_U(rt_dpp<0x124>(TO_U(x))); x = OP(x, y);  /* row_ror:4 */                                          \
    y = FROM_U(rt_dpp<0x128>(TO_U(x))); x = OP(x, y);  /* row_ror:8 */                                          \
    auto a = __builtin_amdgcn_permlane16_swap(TO_U(x), TO_U(x), false, false);                                 \
    x = OP(FROM_U(a[0]), FROM_U(a[1]));                                                                        \
    auto b = __builtin_amdgcn_permlane32_swap(TO_U(x), TO_U(x), false, false);                                 \
    x = OP(FROM_U(b[0]), FROM_U(b[1]));                                                                        \
    return FROM_U((uint32_t)__builtin_amdgcn_readfirstlane((int)TO_U(x))); /* one value for the whole wave */  \
  }
#define RT_ID(x) (x)
__device__ __forceinline__ uint32_t rt_umax(uint32_t a, uint32_t b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ uint32_t rt_umin(uint32_t a, uint32_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ float rt_fadd(float a, float b) { return a + b; }
#define RT_UMAX rt_umax
#define RT_UMIN rt_umin
#define RT_FADD rt_fadd
RT_WAVE_REDUCE(rt_wave_umax, uint32_t, RT_UMAX, RT_ID, RT_ID)
RT_WAVE_REDUCE(rt_wave_umin, uint32_t, RT_UMIN, RT_ID, RT_ID)
RT_WAVE_REDUCE(rt_wave_fmax, float, fmaxf, __float_as_uint, __uint_as_float)
RT_WAVE_REDUCE(rt_wave_fsum, float, RT_FADD, __float_as_uint, __uint_as_float)

// The router tail for ONE token on one wave: fp32 softmax over the row, k rounds of (largest remaining probability, lowest
// expert index among equals), optional renormalisation, cast.  Probabilities are non-negative floats: their bit patterns
// order like the values; a NaN orders above every finite value (as torch.topk orders it) and still yields a valid index.
// v[j] = the logit of expert j * 64 + lane as fp32 (-inf past `experts`)
// WT: the ids are stored write-through (sc1) -- another workgroup of the same launch reads them (the fused router + align)
template <int DT, int PER, bool WT = false>  // PER = ceil(experts / 64)
__device__ __forceinline__ void moe_topk_wave_vals(uint16_t* __restrict__ w_out, int64_t* __restrict__ ids_out, float (&v)[PER],
                                                   int experts, int top_k, int norm, int lane) {
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < PER; ++j) mx = fmaxf(mx, v[j]);
  mx = rt_wave_fmax(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    v[j] = j * 64 + lane < experts ? expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = rt_wave_fsum(sum);
#pragma unroll
  for (int j = 0; j < PER; ++j) v[j] = v[j] / sum;  // the probabilities torch.softmax(dtype=float32) hands to topk
  float picked_w = 0.f;   // lane r keeps selection r
  int picked_e = 0;
  float top_sum = 0.f;
  unsigned taken = 0u;    // bit j: this lane's expert j * 64 + lane is already selected
  for (int r = 0; r < top_k; ++r) {
    uint32_t lb = 0u, le = 0xffffffffu;  // this lane's best remaining (bits, expert); lower j wins a tie
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = j * 64 + lane;
      const uint32_t bits = __float_as_uint(v[j]);
      if (e < experts && !((taken >> j) & 1u) && (le == 0xffffffffu || bits > lb)) {
        lb = bits;
        le = (uint32_t)e;
      }
    }
    const uint32_t best = rt_wave_umax(le == 0xffffffffu ? 0u : lb);
    const uint32_t e = rt_wave_umin((le != 0xffffffffu && lb == best) ? le : 0xffffffffu);
    const float pw = __uint_as_float(best);
    top_sum += pw;
    if (lane == r) {
      picked_w = pw;
      picked_e = (int)e;
    }
    if ((int)(e & 63u) == lane) taken |= 1u << (e >> 6);
  }
  if (lane < top_k) {
    w_out[lane] = from_f32<DT>(norm ? picked_w / top_sum : picked_w);
    if constexpr (WT) __hip_atomic_store(ids_out + lane, (int64_t)picked_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else ids_out[lane] = picked_e;
  }
}

template <int DT, int PER>
__device__ __forceinline__ void moe_topk_wave(uint16_t* __restrict__ w_out, int64_t* __restrict__ ids_out,
                                              const uint16_t* __restrict__ logits, int experts, int top_k, int norm, int lane) {
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = j * 64 + lane;
    v[j] = e < experts ? to_f32<DT>(logits[e]) : -INFINITY;
  }
  moe_topk_wave_vals<DT, PER>(w_out, ids_out, v, experts, top_k, norm, lane);
}

template <int DT, int PER>
__global__ __launch_bounds__(256) void moe_route_topk_kernel(uint16_t* __restrict__ w_out, int64_t* __restrict__ ids_out,
                                                             const uint16_t* __restrict__ logits, int64_t tokens, int experts,
                                                             int64_t l_stride, int top_k, int norm) {
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per token
  if (tok >= tokens) return;
  moe_topk_wave<DT, PER>(w_out + tok * top_k, ids_out + tok * top_k, logits + tok * l_stride, experts, top_k, norm,
                         (int)(threadIdx.x & 63));
}

extern "C" int ll_moe_route_topk(void* weights_out, int64_t* ids_out, const void* logits, int64_t tokens, int experts,
                                 int64_t logits_stride, int top_k, int norm_topk_prob, int dtype, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (tokens < 0 || experts < 1 || experts > 1024 || top_k < 1 || top_k > 64 || top_k > experts || logits_stride < experts)
    return LL_ERR_SHAPE;
  if (tokens == 0) return LL_OK;
  if (!weights_out || !ids_out || !logits) return LL_ERR_ARG;
  const dim3 grid((unsigned)((tokens + 3) / 4));
#define LL_TK(DT, PER)                                                                                             \
  moe_route_topk_kernel<DT, PER><<<grid, 256, 0, (hipStream_t)stream>>>((uint16_t*)weights_out, ids_out, (const uint16_t*)logits, \
                                                                        tokens, experts, logits_stride, top_k, norm_topk_prob)
#define LL_TK_P(DT)                                                                   \
  if (experts <= 64) LL_TK(DT, 1); else if (experts <= 128) LL_TK(DT, 2);             \
  else if (experts <= 256) LL_TK(DT, 4); else if (experts <= 512) LL_TK(DT, 8); else LL_TK(DT, 16)
  if (dtype == LL_F16) { LL_TK_P(LL_F16); } else { LL_TK_P(LL_BF16); }
#undef LL_TK_P
#undef LL_TK
  return LL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------- //
// Router gate in-tree (round 6): lite_llama/models/qwen3_moe.py:102-111 runs ``self.gate(hidden_states)`` -- an unquantised
// fp16 linear [experts, hidden] -- as a library GEMM (Cijk_* in the step's trace: 10.9 us per layer for 64 x 2048 -> 128 on
// Qwen3-30B-A3B, 48 layers).  The gate matrix is 0.5 MB: ONE workgroup cannot pull it through its CU in less than ~12 us (the
// two one-launch routers of round 4), but split along K it is a few KB per workgroup.  Two launches:
//   (1) moe_gate_partials_kernel: grid (experts / 32, ceil(K / 512)), four waves each: a wave multiplies 32 gate rows x 128 k (the
//       MFMA A operand) by the <= 64 token rows (B), both straight from global memory in fragment layout (16 bytes per lane and
//       k-step; a row's eight k-steps are 256 contiguous bytes), 8 x MT MFMA 32x32x16; the four waves meet in LDS in chunk order
//       and the fp32 tile leaves as plane [slice][token][expert];
//   (2) the router tail above, reading a token's logits as the sum of the planes in slice order, rounded ONCE to the
//       activation dtype -- the value the reference's fp16 GEMM stores -- before the fp32 softmax.
// ---------------------------------------------------------------------------------- //
typedef __bf16 moe_bf16x8 __attribute__((ext_vector_type(8)));
template <int DT>
__device__ __forceinline__ f32x16 moe_mfma32(const Q4& a, const Q4& b, f32x16 c) {
  if constexpr (DT == LL_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(moe_bf16x8, a), __builtin_bit_cast(moe_bf16x8, b), c, 0, 0, 0);
}

template <int DT, int MT>
__global__ __launch_bounds__(256) void moe_gate_partials_kernel(float* __restrict__ planes, const uint16_t* __restrict__ x,
                                                                const uint16_t* __restrict__ w, int tokens, int experts, int chunks,
                                                                int64_t x_stride, int64_t w_stride) {
  __shared__ __attribute__((aligned(16))) f32x4 red[3][MT][4][64];
  const int rg = (int)blockIdx.x, s = (int)blockIdx.y;
  const int lane = (int)(threadIdx.x & 63), nl = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = s * 4 + wv;  // this wave's 128-k chunk of the workgroup's 512-k slice
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
  if (c < chunks) {
    const uint16_t* wrow = w + (int64_t)(rg * 32 + nl) * w_stride + c * 128 + 8 * h;
    Q4 a[8], b[MT][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const Q4*>(wrow + 16 * j);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int tok = mt * 32 + nl;
      if (tok >= tokens) tok = tokens - 1;  // rows >= tokens feed only unstored outputs
      const uint16_t* xrow = x + (int64_t)tok * x_stride + c * 128 + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[mt][j] = *reinterpret_cast<const Q4*>(xrow + 16 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = moe_mfma32<DT>(a[j], b[mt][j], acc[mt]);
  }
  if (wv > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        red[wv - 1][mt][g][lane] = f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
  }
  __syncthreads();
  if (wv > 0) return;
  // the four chunks of the slice meet in chunk order; a lane holds, for token nl (+ 32 mt), the gate rows 8 g + 4 h .. + 3 of the 32
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 r0 = red[0][mt][g][lane], r1 = red[1][mt][g][lane], r2 = red[2][mt][g][lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] = ((acc[mt][4 * g + e] + r0[e]) + r1[e]) + r2[e];
    }
    const int tok = mt * 32 + nl;
    if (tok < tokens) {
      float* dst = planes + ((int64_t)s * tokens + tok) * experts + rg * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
    }
  }
}

// the tail over planes: logits[e] = DT(sum_s planes[s][tok][e]) (slice order), then moe_topk_wave's arithmetic
template <int DT, int PER>
__global__ __launch_bounds__(256) void moe_route_topk_planes_kernel(uint16_t* __restrict__ w_out, int64_t* __restrict__ ids_out,
                                                                    const float* __restrict__ planes, int slices, int64_t tokens,
                                                                    int experts, int top_k, int norm) {
  const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int64_t tok = (int64_t)blockIdx.x * 4 + wv;  // one wave per token
  if (tok >= tokens) return;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = j * 64 + lane;
    float acc = 0.f;
    if (e < experts) {
      const float* src = planes + tok * experts + e;
      for (int s0 = 0; s0 < slices; s0 += 8) {  // eight loads in flight, summed in slice order (absent slices add an exact zero)
        float part[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) part[i] = s0 + i < slices ? src[(int64_t)(s0 + i) * tokens * experts] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += part[i];
      }
    }
    v[j] = e < experts ? to_f32<DT>(from_f32<DT>(acc)) : -INFINITY;  // the 16-bit logit the reference's GEMM stores
  }
  moe_topk_wave_vals<DT, PER>(w_out + tok * top_k, ids_out + tok * top_k, v, experts, top_k, norm, lane);
}

// The tail over planes AND moe_align_block_size in one launch: one wave per token as above (blockDim / 64 tokens per workgroup), ids
// written through; the workgroup that arrives LAST at the launch's counter (relaxed agent-scope fetch-add after its own stores have
// drained; it leaves the counter at zero for the next launch) runs the decode-sized align body over all tokens' ids with coherent
// loads.  No workgroup waits for another one (no residency assumption).  blockDim >= max(slots, experts).
template <int DT, int PER>
__global__ __launch_bounds__(1024) void moe_route_align_kernel(uint16_t* __restrict__ w_out, int64_t* __restrict__ ids_out,
                                                               const float* __restrict__ planes, int slices, int tokens, int experts,
                                                               int top_k, int norm, int block_size, int32_t* __restrict__ sorted_ids,
                                                               int32_t* __restrict__ expert_ids, int32_t* __restrict__ num_post,
                                                               int max_padded, int max_blocks, int32_t* __restrict__ counter) {
  extern __shared__ int sm[];
  __shared__ int is_last;
  const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), nw = (int)(blockDim.x >> 6);
  const int tok = (int)blockIdx.x * nw + wv;
  if (tok < tokens) {
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = j * 64 + lane;
      float acc = 0.f;
      if (e < experts) {
        const float* src = planes + (int64_t)tok * experts + e;
        for (int s0 = 0; s0 < slices; s0 += 8) {
          float part[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) part[i] = s0 + i < slices ? src[(int64_t)(s0 + i) * tokens * experts] : 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += part[i];
        }
      }
      v[j] = e < experts ? to_f32<DT>(from_f32<DT>(acc)) : -INFINITY;
    }
    moe_topk_wave_vals<DT, PER, true>(w_out + (int64_t)tok * top_k, ids_out + (int64_t)tok * top_k, v, experts, top_k, norm, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's ids are out before the counter moves
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = old == (int)gridDim.x - 1;
    if (is_last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  moe_align_small_body<true>(ids_out, LL_I64, tokens * top_k, experts, block_size, sorted_ids, expert_ids, num_post, max_padded,
                             max_blocks, sm);
}

// Shapes the in-tree router serves: decode batches (<= 64 tokens), experts a multiple of 32 (<= 1024), hidden a multiple of 128
// with at most 64 slices.  1 / 0.
extern "C" int ll_moe_router_supported(int64_t tokens, int experts, int64_t hidden) {
  return tokens >= 1 && tokens <= 64 && experts >= 32 && experts <= 1024 && experts % 32 == 0 && hidden >= 128 && hidden % 128 == 0 &&
                 hidden / 512 <= 64 ? 1 : 0;
}
extern "C" int64_t ll_moe_router_workspace_floats(int64_t tokens, int experts, int64_t hidden) {
  return ll_moe_router_supported(tokens, experts, hidden) ? ((hidden + 511) / 512) * tokens * experts : 0;
}

// gate GEMM + softmax + top-k + renormalise + cast; planes: ll_moe_router_workspace_floats fp32 of scratch (no zeroing needed)
extern "C" int ll_moe_router(void* weights_out, int64_t* ids_out, const void* x, const void* gate_w, float* planes, int64_t tokens,
                             int experts, int64_t hidden, int64_t x_stride, int64_t w_stride, int top_k, int norm_topk_prob,
                             int dtype, int align_block, int32_t* sorted_ids, int32_t* expert_ids, int32_t* num_post,
                             int32_t* counter, void* stream) {
  if (dtype != LL_F16 && dtype != LL_BF16) return LL_ERR_DTYPE;
  if (!ll_moe_router_supported(tokens, experts, hidden) || top_k < 1 || top_k > 64 || top_k > experts || x_stride < hidden ||
      w_stride < hidden || x_stride % 8 != 0 || w_stride % 8 != 0)
    return LL_ERR_SHAPE;
  if (!weights_out || !ids_out || !x || !gate_w || !planes || !ll_aligned16(x) || !ll_aligned16(gate_w) || !ll_aligned16(planes))
    return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int slices = (int)((hidden + 511) / 512);
  const dim3 g1((unsigned)(experts / 32), (unsigned)slices);
#define LL_GP(DT, MT)                                                                                                        \
  moe_gate_partials_kernel<DT, MT><<<g1, 256, 0, st>>>(planes, (const uint16_t*)x, (const uint16_t*)gate_w, (int)tokens, experts, \
                                                       (int)(hidden / 128), x_stride, w_stride)
  if (dtype == LL_F16) { if (tokens > 32) LL_GP(LL_F16, 2); else LL_GP(LL_F16, 1); }
  else { if (tokens > 32) LL_GP(LL_BF16, 2); else LL_GP(LL_BF16, 1); }
#undef LL_GP
  if (hipGetLastError() != hipSuccess) return LL_ERR_LAUNCH;
  if (align_block > 0) {
    // + moe_align_block_size(ids, align_block, experts) of the same launch when the decode-sized align body fits one workgroup
    if (!sorted_ids || !expert_ids || !num_post || !counter) return LL_ERR_ARG;
    const int64_t slots = tokens * top_k;
    const int max_padded = (int)slots + experts * (align_block - 1);
    const int max_blocks = (max_padded + align_block - 1) / align_block;
    int threads = (int)((slots > experts ? slots : experts) + 63) / 64 * 64;
    if (threads < 256) threads = 256;
    const size_t lds = moe_align_small_lds(threads, experts);
    static const bool fuse_off = getenv("LL_MOE_ROUTER_NO_FUSED_ALIGN") != nullptr;  // A/B knob, read once
    if (!fuse_off && threads <= 1024 && lds <= 48 * 1024) {
      const dim3 g3((unsigned)((tokens + threads / 64 - 1) / (threads / 64)));
#define LL_RA(DT, PER)                                                                                                             \
  moe_route_align_kernel<DT, PER><<<g3, threads, lds, st>>>((uint16_t*)weights_out, ids_out, planes, slices, (int)tokens, experts, \
                                                            top_k, norm_topk_prob, align_block, sorted_ids, expert_ids, num_post,  \
                                                            max_padded, max_blocks, counter)
#define LL_RA_P(DT)                                                                   \
  if (experts <= 64) LL_RA(DT, 1); else if (experts <= 128) LL_RA(DT, 2);             \
  else if (experts <= 256) LL_RA(DT, 4); else if (experts <= 512) LL_RA(DT, 8); else LL_RA(DT, 16)
      if (dtype == LL_F16) { LL_RA_P(LL_F16); } else { LL_RA_P(LL_BF16); }
#undef LL_RA_P
#undef LL_RA
      return LL_LAUNCH_CHECK();
    }
  }
  const dim3 g2((unsigned)((tokens + 3) / 4));
#define LL_TKP(DT, PER)                                                                                                        \
  moe_route_topk_planes_kernel<DT, PER><<<g2, 256, 0, st>>>((uint16_t*)weights_out, ids_out, planes, slices, tokens, experts, top_k, \
                                                            norm_topk_prob)
#define LL_TKP_P(DT)                                                                    \
  if (experts <= 64) LL_TKP(DT, 1); else if (experts <= 128) LL_TKP(DT, 2);             \
  else if (experts <= 256) LL_TKP(DT, 4); else if (experts <= 512) LL_TKP(DT, 8); else LL_TKP(DT, 16)
  if (dtype == LL_F16) { LL_TKP_P(LL_F16); } else { LL_TKP_P(LL_BF16); }
#undef LL_TKP_P
#undef LL_TKP
  if (hipGetLastError() != hipSuccess) return LL_ERR_LAUNCH;
  if (align_block > 0)
    return ll_moe_align_block_size(ids_out, LL_I64, tokens * top_k, experts, align_block, sorted_ids, expert_ids, num_post, stream);
  return LL_OK;
}
