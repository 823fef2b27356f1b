"""TEST INFRASTRUCTURE ONLY -- plain-python model of the block-granular KV pool of csrc/kv_paged.hip.

PARITY UNPINNED against the reference: it has no block-granular pool (the TODO at
lite_llama/executor/kv_cache_manager.py:211).  What the device code is held to: (1) this model, exactly (tables, select
rows, stack, counters, error flags over random admit / append / release sequences), and (2) the reference's OBSERVABLE
behaviour -- decode over rows handed out by the paged pool produces the tokens that decode over the reference's
token-granular rows produces (tests/test_paged_kv_gpu.py)."""

from __future__ import annotations

import numpy as np


class PagedPoolModel:
    def __init__(self, num_blocks, block_size, max_requests, max_seq_len):
        self.nb, self.bs = num_blocks, block_size
        self.bpr = (max_seq_len + block_size - 1) // block_size
        self.stack = [num_blocks - 2 - i for i in range(num_blocks - 1)]   # list end = ... top is index free-1
        self.free = num_blocks - 1
        self.err = 0
        self.block_table = np.zeros((max_requests, self.bpr), dtype=np.int32)
        self.req_blocks = np.zeros(max_requests, dtype=np.int32)
        self.token_table = np.zeros((max_requests, max_seq_len), dtype=np.int32)

    def extend(self, req_idx, lens, grid_len, from_end):
        needs, over = [], False
        for r, ln in zip(req_idx, lens):
            want = (ln + self.bs - 1) // self.bs if ln > 0 else 0
            if want > self.bpr:
                want, over = self.bpr, True
            needs.append(max(want - int(self.req_blocks[r]), 0))
        if sum(needs) > self.free:
            self.err |= 1
        else:
            if over:
                self.err |= 2
            prefix = 0
            for r, need in zip(req_idx, needs):
                held = int(self.req_blocks[r])
                for j in range(need):
                    self.block_table[r, held + j] = self.stack[self.free - 1 - prefix - j]
                self.req_blocks[r] = held + need
                prefix += need
            self.free -= sum(needs)
        select = np.zeros(len(req_idx) * grid_len, dtype=np.int32)
        for i, (r, ln) in enumerate(zip(req_idx, lens)):
            for g in range(grid_len):
                p = ln - grid_len + g if from_end else g
                row = (self.nb - 1) * self.bs + (p % self.bs if p >= 0 else 0)
                if 0 <= p < ln and p // self.bs < self.req_blocks[r]:
                    row = int(self.block_table[r, p // self.bs]) * self.bs + p % self.bs
                    self.token_table[r, p] = row
                select[i * grid_len + g] = row
        return select

    def release(self, req_idx):
        for r in req_idx:
            for j in range(int(self.req_blocks[r])):
                if self.free < len(self.stack):
                    self.stack[self.free] = int(self.block_table[r, j])
                self.free += 1
            self.req_blocks[r] = 0
