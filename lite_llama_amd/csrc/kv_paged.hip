// Block-granular KV paging on the device (second half).  The reference's pool is token-granular and
// carries the TODO "reshape into [blocks, block_size, ...] to support PagedAttention"
// (lite_llama/executor/kv_cache_manager.py:211); its kernels read a PER-TOKEN table
// (b_req_tokens_table[req][pos] -> pool row, executor/model_runner.py:153-218).  Here the pool is handed out in blocks
// of ``block_size`` consecutive rows -- a request holds ceil(len / block_size) blocks, listed in a block table -- and the
// per-token table the kernels consume is DERIVED from it on the device (row = block * block_size + pos % block_size):
// the attention / KV-write kernels and their results are unchanged, allocation work drops by block_size x, a
// request's rows are contiguous in runs of block_size, and nothing is read back by the host (the free blocks live in a
// device stack; admission, the per-step append and release are stream-ordered launches that can sit inside the
// captured decode step).
//
// State (all int32, device):  free_stack[num_blocks - 1] block ids, state[0] = number of free blocks (the stack
// top), state[1] = error flags (1: out of blocks, 2: a request outgrew its block-table row); block_table
// [max_reqs][bt_stride]; req_blocks[max_reqs] = blocks held.  Block num_blocks - 1 is never handed out: it is the
// JUNK block, the rows that padded prefill positions and refused requests write to (the kernels always get valid
// rows).  Pops come off the top in request order (an exclusive scan over the batch), so the outcome is deterministic:
// a fresh pool hands out blocks 0, 1, 2, ...
#include "common.h"

namespace {

constexpr int kMaxBatch = 1024;  // one workgroup scans the batch

// exclusive scan of one int per thread over a 1024-thread workgroup (lds: 16 wave sums + 64 scanned); returns
// (exclusive prefix, total)
__device__ __forceinline__ void block_scan(int v, int* lds, int& prefix, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  if (wave == 0) {
    int w = lane < (int)(blockDim.x >> 6) ? lds[lane] : 0;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int y = __shfl_up(w, off, 64);
      if (lane >= off) w += y;
    }
    lds[16 + lane] = w;  // inclusive wave totals
  }
  __syncthreads();
  const int before = wave == 0 ? 0 : lds[16 + wave - 1];
  prefix = before + x - v;
  total = lds[16 + (blockDim.x >> 6) - 1];
  __syncthreads();
}

__global__ void paged_reset_kernel(int32_t* free_stack, int32_t* state, int32_t* req_blocks, int64_t num_blocks,
                                   int64_t max_reqs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t usable = num_blocks - 1;
  if (i < usable) free_stack[i] = (int32_t)(usable - 1 - i);  // top of the stack = block 0
  if (i < max_reqs) req_blocks[i] = 0;
  if (i == 0) {
    state[0] = (int32_t)usable;
    state[1] = 0;
  }
}

// Make request req_idx[i] hold ceil(want_len[i] / block_size) blocks (never shrinks).  All-or-nothing per call: if the
// stack is short nothing is popped and error bit 1 is set.
__global__ __launch_bounds__(kMaxBatch) void paged_reserve_kernel(
    const int32_t* __restrict__ free_stack, int32_t* __restrict__ state, int32_t* __restrict__ block_table,
    int64_t bt_stride, int32_t* __restrict__ req_blocks, const int32_t* __restrict__ req_idx,
    const int32_t* __restrict__ want_len, int len_bias, int n, int block_size, int max_blocks) {
  __shared__ int lds[96];
  const int i = threadIdx.x;
  int req = 0, held = 0, need = 0;
  bool overflow = false;
  if (i < n) {
    req = req_idx[i];
    held = req_blocks[req];
    const int len = want_len[i] + len_bias;
    int want = len > 0 ? (len + block_size - 1) / block_size : 0;
    if (want > max_blocks) {
      want = max_blocks;
      overflow = true;
    }
    need = want > held ? want - held : 0;
  }
  int prefix, total;
  block_scan(need, lds, prefix, total);
  const int top = state[0];
  __syncthreads();  // everyone has read the top before it moves
  if (total > top) {
    if (i == 0) atomicOr(&state[1], 1);
    return;
  }
  if (overflow) atomicOr(&state[1], 2);
  for (int j = 0; j < need; ++j) block_table[(int64_t)req * bt_stride + held + j] = free_stack[top - 1 - prefix - j];
  if (need > 0) req_blocks[req] = held + need;
  if (i == 0) state[0] = top - total;
}

// Caller contract (not checked on the device: the entry points carry no request-table size): request indices are inside
// the tables and no request is listed twice in one call.
// Rows of positions [p_lo, p_hi) of every request on a [n][grid_len] grid: token_table[req][p] = row and
// select_out[i * grid_len + (p - p_base)] = row; positions the request holds no block for (pads, refused requests)
// get junk-block rows in select_out and leave the token table alone.
__global__ void paged_rows_kernel(const int32_t* __restrict__ block_table, int64_t bt_stride,
                                  const int32_t* __restrict__ req_blocks, const int32_t* __restrict__ req_idx,
                                  const int32_t* __restrict__ lens, int len_bias, int n, int block_size, int grid_len,
                                  int from_end, int32_t* __restrict__ token_table, int64_t tt_stride,
                                  int32_t* __restrict__ select_out, int64_t num_blocks, int32_t* __restrict__ state) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * grid_len) return;
  const int i = (int)(idx / grid_len), g = (int)(idx - (int64_t)i * grid_len);
  const int req = req_idx[i];
  const int len = lens[i] + len_bias;
  // from_end: the grid holds the LAST grid_len positions of the request (the decode step: grid_len = 1 -> position len - 1)
  const int p = from_end ? len - grid_len + g : g;
  const int64_t junk = (num_blocks - 1) * block_size + (p >= 0 ? p % block_size : 0);
  int64_t row = junk;
  if (p >= 0 && p < len && p >= tt_stride) {
    atomicOr(&state[1], 2);  // beyond the request's token-table row (max_seq_len not a multiple of block_size): junk row, flagged
  } else if (p >= 0 && p < len) {
    const int b = p / block_size;
    if (b < req_blocks[req]) {
      row = (int64_t)block_table[(int64_t)req * bt_stride + b] * block_size + p % block_size;
      token_table[(int64_t)req * tt_stride + p] = (int32_t)row;
    }
  }
  select_out[idx] = (int32_t)row;
}

// Return every block of the listed requests to the stack (request order, block order); req_blocks -> 0.
__global__ __launch_bounds__(kMaxBatch) void paged_release_kernel(int32_t* __restrict__ free_stack, int32_t* __restrict__ state,
                                                                  const int32_t* __restrict__ block_table, int64_t bt_stride,
                                                                  int32_t* __restrict__ req_blocks,
                                                                  const int32_t* __restrict__ req_idx, int n) {
  __shared__ int lds[96];
  const int i = threadIdx.x;
  int req = 0, held = 0;
  if (i < n) {
    req = req_idx[i];
    held = req_blocks[req];
  }
  int prefix, total;
  block_scan(held, lds, prefix, total);
  const int top = state[0];
  __syncthreads();
  for (int j = 0; j < held; ++j) free_stack[top + prefix + j] = block_table[(int64_t)req * bt_stride + j];
  if (i < n) req_blocks[req] = 0;
  if (i == 0) state[0] = top + total;
}

}  // namespace

extern "C" int ll_kv_paged_reset(int32_t* free_stack, int32_t* state, int32_t* req_blocks, int64_t num_blocks,
                                 int64_t max_reqs, void* stream) {
  if (num_blocks < 2 || max_reqs < 1) return LL_ERR_SHAPE;
  if (!free_stack || !state || !req_blocks) return LL_ERR_ARG;
  const int64_t n = num_blocks > max_reqs ? num_blocks : max_reqs;
  paged_reset_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(free_stack, state, req_blocks,
                                                                                        num_blocks, max_reqs);
  return LL_LAUNCH_CHECK();
}

// lens + len_bias = the length every request must be able to hold after the call; rows are produced for the grid
// [n][grid_len]: from_end = 0 -> positions 0 .. grid_len - 1 (prefill, pads included), from_end = 1 -> the last
// grid_len positions (decode append: grid_len = 1).
extern "C" int ll_kv_paged_extend(int32_t* free_stack, int32_t* state, int32_t* block_table, int64_t bt_stride,
                                  int32_t* req_blocks, const int32_t* req_idx, const int32_t* lens, int len_bias,
                                  int64_t n, int block_size, int64_t grid_len, int from_end, int32_t* token_table,
                                  int64_t tt_stride, int32_t* select_out, int64_t num_blocks, void* stream) {
  if (n < 0 || n > kMaxBatch || block_size < 1 || grid_len < 1 || num_blocks < 2 || bt_stride < 1) return LL_ERR_SHAPE;
  if (n == 0) return LL_OK;
  if (!free_stack || !state || !block_table || !req_blocks || !req_idx || !lens || !token_table || !select_out)
    return LL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  paged_reserve_kernel<<<1, kMaxBatch, 0, st>>>(free_stack, state, block_table, bt_stride, req_blocks, req_idx, lens,
                                               len_bias, (int)n, block_size, (int)bt_stride);
  const int64_t total = n * grid_len;
  paged_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(
      block_table, bt_stride, req_blocks, req_idx, lens, len_bias, (int)n, block_size, (int)grid_len, from_end,
      token_table, tt_stride, select_out, num_blocks, state);
  return LL_LAUNCH_CHECK();
}

extern "C" int ll_kv_paged_release(int32_t* free_stack, int32_t* state, const int32_t* block_table, int64_t bt_stride,
                                   int32_t* req_blocks, const int32_t* req_idx, int64_t n, void* stream) {
  if (n < 0 || n > kMaxBatch || bt_stride < 1) return LL_ERR_SHAPE;
  if (n == 0) return LL_OK;
  if (!free_stack || !state || !block_table || !req_blocks || !req_idx) return LL_ERR_ARG;
  paged_release_kernel<<<1, kMaxBatch, 0, (hipStream_t)stream>>>(free_stack, state, block_table, bt_stride, req_blocks,
                                                                req_idx, (int)n);
  return LL_LAUNCH_CHECK();
}
