"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's KV row allocator
(lite_llama/executor/kv_cache_manager.py:158-373): use counts, the free-row counter, the
append-only cursor, first-contiguous-run search and the scattered fallback.

Pinned by tests/golden/kv_alloc_sequence.npz (an op sequence recorded from the reference class by
tests/golden/gen_golden_kv_alloc.py).  Only tests may import it.
"""

from __future__ import annotations

import numpy as np


class OracleKVAllocator:
    def __init__(self, rows: int):
        self.n = rows
        self.state = np.zeros(rows, dtype=np.int32)
        self.free = rows
        self.cursor, self.cursor_exact = 0, True

    # :219-231
    def alloc_kvcache(self, need):
        if need > self.free:
            return None
        sel = np.nonzero(self.state == 0)[0][:need]
        self.add_ref(sel)
        return sel.astype(np.int64)

    # :234-267
    def alloc_contiguous_kvcache(self, need):
        if need > self.free:
            return None
        fr = np.nonzero(self.state == 0)[0]
        if need <= fr.size and need > 0:
            starts, ends = fr[: fr.size - need + 1], fr[need - 1:]
            hit = np.nonzero(ends - starts == need - 1)[0]
            if hit.size:
                s = int(starts[hit[0]])
                sel = np.arange(s, s + need)
                self.add_ref(sel)
                return sel.astype(np.int64), s, s + need
        return None

    # :270-299
    def alloc_kvcache_index(self, need):
        if self.cursor_exact and self.cursor + need <= self.n:
            s = self.cursor
            self.state[s:s + need] += 1
            self.cursor += need
            self.free -= need
            return np.arange(s, s + need, dtype=np.int32)
        got = self.alloc_contiguous_kvcache(need)
        if got is not None:
            return got[0].astype(np.int32)
        sel = self.alloc_kvcache(need)
        return None if sel is None else sel.astype(np.int32)

    # :302-314
    def add_ref(self, idx):
        st = self.state[idx]
        self.free -= len(st) - int(np.count_nonzero(st))
        self.state[idx] += 1

    # :317-333
    def release_ref(self, idx):
        self.cursor_exact = False
        u, c = np.unique(np.asarray(idx), return_counts=True)
        self.state[u] -= c.astype(np.int32)
        st = self.state[u]
        self.free += len(st) - int(np.count_nonzero(st))

    # :336-353
    def claim(self, rows):
        if rows > self.free:
            raise ValueError("cannot claim")
        self.state[:rows] += 1
        self.free -= rows
        self.cursor = max(self.cursor, rows)

    def free_all(self):
        self.free = self.n
        self.state[:] = 0
        self.cursor, self.cursor_exact = 0, True
