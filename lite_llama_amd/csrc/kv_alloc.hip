// Device-side KV row allocator.  The reference's general path
// (executor/kv_cache_manager.py:219-267: alloc_kvcache / alloc_contiguous_kvcache) finds free rows
// with ``nonzero`` over the whole use-count vector plus two ``.item()`` reads -- three device
// synchronisations per call.  Here the same answers are produced on the device with no read-back:
//   contiguous-first:  the FIRST run of ``need`` consecutive free rows, else
//   scattered:         the first ``need`` free rows in ascending order,
// rows are marked used (count += 1) and the free-row counter is debited, all stream-ordered.
// Integer work: results are exactly the reference's (pinned by recorded op sequences + the oracle).
//
// Three launches:  scan (one wave per 4096-row chunk: 64 ballot words, per-chunk summary)
//                  -> decide (one wave walks the chunk summaries: cross-chunk runs, prefix of free counts)
//                  -> fill (writes the row ids, bumps the counts).
#include "common.h"

namespace {

constexpr int kChunk = 4096;  // rows per wave: 64 words of 64 rows

struct ChunkSummary {
  int32_t free_rows;  // zero-count rows in the chunk
  int32_t lead;       // leading free rows (kChunk when all are free)
  int32_t trail;      // trailing free rows
  int32_t first_run;  // offset of the first run >= need that lies inside the chunk, or -1
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// the 64 free-masks of one chunk: lane j ends up holding the mask of rows [base + 64 j, base + 64 j + 64)
__device__ __forceinline__ uint64_t chunk_masks(const int32_t* __restrict__ state, int64_t base, int64_t n) {
  const int lane = lane_id();
  uint64_t mine = 0;
#pragma unroll 8
  for (int j = 0; j < 64; ++j) {
    const int64_t row = base + (int64_t)j * 64 + lane;
    const bool is_free = row < n && state[row] == 0;
    const uint64_t m = __ballot(is_free);
    if (lane == j) mine = m;
  }
  return mine;
}

__global__ __launch_bounds__(256) void kv_scan_kernel(const int32_t* __restrict__ state, int64_t n, int64_t need,
                                                      ChunkSummary* __restrict__ summ, int64_t nchunks) {
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  const int lane = lane_id();
  const uint64_t mask = chunk_masks(state, chunk * kChunk, n);
  const bool all = mask == ~0ull;
  const int lead = all ? 64 : __builtin_ctzll(~mask);
  const int trail = all ? 64 : __builtin_clzll(~mask);
  // first position inside this word where ``need`` consecutive bits are set (need <= 64 only)
  int inside = -1;
  if (need <= 64) {
    uint64_t r = mask;
    int len = 1;
    while (len < (int)need) {
      const int s = min(len, (int)need - len);
      r &= r >> s;
      len += s;
    }
    inside = r ? __builtin_ctzll(r) : -1;
  }
  int cnt = __builtin_popcountll(mask);
  for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off);
  // walk the 64 words in order (uniform across the wave)
  int64_t run = 0, run_start = 0, best = -1;
  int clead = 0;
  bool lead_done = false;
  for (int j = 0; j < 64; ++j) {
    const int l = __shfl(lead, j), t = __shfl(trail, j), in = __shfl(inside, j);
    const bool a = __shfl((int)all, j) != 0;
    const int64_t wbase = (int64_t)j * 64;
    if (best < 0) {
      if (run + l >= need && run + l > 0) best = run > 0 ? run_start : wbase;
      else if (in >= 0) best = wbase + in;
    }
    if (!lead_done) {
      clead += l;
      lead_done = !a;
    }
    if (a) {
      if (run == 0) run_start = wbase;
      run += 64;
    } else {
      run = t;
      run_start = wbase + 64 - t;
    }
  }
  if (lane == 0) summ[chunk] = ChunkSummary{cnt, clead, (int32_t)run, (int32_t)best};
}

// decision[0] = mode (0 none, 1 contiguous, 2 scattered), decision[1] = first row of the run (mode 1)
__global__ __launch_bounds__(64) void kv_decide_kernel(const ChunkSummary* __restrict__ summ, int64_t nchunks,
                                                       int64_t need, int contiguous_first, int64_t* __restrict__ prefix,
                                                       int64_t* __restrict__ decision, int64_t* __restrict__ free_rows) {
  const int lane = lane_id();
  int64_t run = 0, run_start = 0, found = -1, pre = 0;
  for (int64_t c0 = 0; c0 < nchunks; c0 += 64) {
    const int64_t c = c0 + lane;
    ChunkSummary s{0, 0, 0, -1};
    if (c < nchunks) s = summ[c];
    int64_t my_prefix = 0;
    const int lim = (int)min((int64_t)64, nchunks - c0);
    for (int j = 0; j < lim; ++j) {
      const int cnt = __shfl(s.free_rows, j), l = __shfl(s.lead, j), t = __shfl(s.trail, j), b = __shfl(s.first_run, j);
      const int64_t base = (c0 + j) * kChunk;
      if (lane == j) my_prefix = pre;
      pre += cnt;
      if (found < 0) {
        if (run + l >= need && run + l > 0) found = run > 0 ? run_start : base;
        else if (b >= 0) found = base + b;
      }
      if (l == kChunk) {
        if (run == 0) run_start = base;
        run += kChunk;
      } else {
        run = t;
        run_start = base + kChunk - t;
      }
    }
    if (c < nchunks) prefix[c] = my_prefix;
  }
  if (lane == 0) {
    int64_t mode = 0;
    if (need > 0 && pre >= need) mode = (contiguous_first && found >= 0) ? 1 : 2;
    if (contiguous_first == 2 && found < 0) mode = 0;  // contiguous ONLY (alloc_contiguous_kvcache)
    decision[0] = mode;
    decision[1] = mode == 1 ? found : -1;
    if (mode) *free_rows -= need;
  }
}

__global__ __launch_bounds__(256) void kv_fill_kernel(int32_t* __restrict__ state, int64_t n, int64_t need,
                                                      const int64_t* __restrict__ prefix,
                                                      const int64_t* __restrict__ decision, int32_t* __restrict__ out,
                                                      int64_t nchunks) {
  const int64_t mode = decision[0];
  if (mode == 1) {
    const int64_t start = decision[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < need; i += (int64_t)gridDim.x * 256) {
      out[i] = (int32_t)(start + i);
      state[start + i] += 1;
    }
    return;
  }
  if (mode != 2) return;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  int64_t rank0 = prefix[chunk];
  if (rank0 >= need) return;
  const int lane = lane_id();
  const int64_t base = chunk * kChunk;
  for (int j = 0; j < 64 && rank0 < need; ++j) {
    const int64_t row = base + (int64_t)j * 64 + lane;
    const bool is_free = row < n && state[row] == 0;
    const uint64_t m = __ballot(is_free);
    const int64_t rank = rank0 + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (is_free && rank < need) {
      out[rank] = (int32_t)row;
      state[row] = 1;
    }
    rank0 += __builtin_popcountll(m);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kv_ref_kernel(int32_t* __restrict__ state, const T* __restrict__ index,
                                                     int64_t count, int delta, int64_t* __restrict__ free_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int64_t row = (int64_t)index[i];
  if (delta > 0) {
    if (atomicAdd(&state[row], 1) == 0) atomicAdd((unsigned long long*)free_rows, (unsigned long long)-1ll);
  } else {
    if (atomicSub(&state[row], 1) == 1) atomicAdd((unsigned long long*)free_rows, 1ull);
  }
}

}  // namespace

extern "C" int64_t ll_kv_alloc_scratch_bytes(int64_t n_rows) {
  const int64_t nchunks = (n_rows + kChunk - 1) / kChunk;
  return nchunks * (int64_t)(sizeof(ChunkSummary) + sizeof(int64_t)) + 64;
}

extern "C" int ll_kv_alloc(int32_t* state, int64_t n_rows, int64_t need, int contiguous_first, int32_t* out_index,
                           void* scratch, int64_t* decision, int64_t* free_rows, void* stream) {
  if (!state || !out_index || !scratch || !decision || !free_rows) return LL_ERR_ARG;
  if (n_rows <= 0 || need < 0 || need > n_rows || n_rows > (int64_t)INT32_MAX) return LL_ERR_SHAPE;
  if (contiguous_first < 0 || contiguous_first > 2) return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nchunks = (n_rows + kChunk - 1) / kChunk;
  int64_t* prefix = (int64_t*)scratch;
  ChunkSummary* summ = (ChunkSummary*)(prefix + nchunks);
  const unsigned blocks = (unsigned)((nchunks + 3) / 4);
  kv_scan_kernel<<<dim3(blocks), 256, 0, st>>>(state, n_rows, need, summ, nchunks);
  kv_decide_kernel<<<dim3(1), 64, 0, st>>>(summ, nchunks, need, contiguous_first, prefix, decision, free_rows);
  kv_fill_kernel<<<dim3(blocks), 256, 0, st>>>(state, n_rows, need, prefix, decision, out_index, nchunks);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_kv_ref_update(int32_t* state, int64_t n_rows, const void* index, int64_t count, int idx_width,
                                int delta, int64_t* free_rows, void* stream) {
  if (!state || !free_rows || (count > 0 && !index)) return LL_ERR_ARG;
  if (idx_width != LL_I32 && idx_width != LL_I64) return LL_ERR_DTYPE;
  if (count < 0 || n_rows <= 0 || (delta != 1 && delta != -1)) return LL_ERR_SHAPE;
  if (count == 0) return LL_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((count + 255) / 256));
  if (idx_width == LL_I32)
    kv_ref_kernel<int32_t><<<grid, 256, 0, st>>>(state, (const int32_t*)index, count, delta, free_rows);
  else
    kv_ref_kernel<int64_t><<<grid, 256, 0, st>>>(state, (const int64_t*)index, count, delta, free_rows);
  return LL_LAUNCH_CHECK();
}
