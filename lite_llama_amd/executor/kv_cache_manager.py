"""Token-granular KV pool with reference-counted rows -- the counterpart of
lite_llama/executor/kv_cache_manager.py:140-373 (same constructor, attributes and methods, same
answers), with the general allocation path moved onto the device.

What the reference does per general allocation: ``nonzero`` over the use-count vector, a windowed
comparison and two ``.item()`` reads (:219-267) -- three host synchronisations; ``add_ref`` /
``release_ref`` read a count back as well (:302-333).  Here the search (``ll_kv_alloc``) and the
count updates (``ll_kv_ref_update``) are stream-ordered launches; the number of free rows lives on
the device and the host keeps a lower bound of it, which is all an admission check needs.  The host
only reads the device counter when (a) someone asks for ``can_use_mem_size`` while the bound is
not exact, or (b) a request does not fit the bound.  ``alloc_contiguous_kvcache`` keeps the
reference's return type (python ints ``start`` / ``end``) and therefore still reads back; the
hot entry ``alloc_kvcache_index`` does not.

Kept as in the reference: only ``release_ref`` retires the append-only cursor, so calling the two
general allocators directly while the cursor is still exact hands rows out that the cursor will hand
out again (the reference's engines never mix the two).

Difference kept on purpose: ``add_ref`` with a row named twice counts it twice (the reference's
``index_put`` counts it once but debits the free counter twice); callers never do that.
"""

from __future__ import annotations

import torch

from .. import _lib as L


class KVCacheManager:
    def __init__(self, num_layers, num_kv_heads, head_dim, gpu_num_blocks, block_size=1, dtype=torch.float16,
                 device="cuda"):
        if block_size != 1:
            raise ValueError("only block_size 1 (token attention) is implemented, as in the reference")
        self.num_layers, self.num_kv_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.gpu_num_blocks, self.block_size = gpu_num_blocks, block_size
        self.max_num_tokens = gpu_num_blocks * block_size
        self.dtype, self.device = dtype, device
        n = self.max_num_tokens
        self.kv_mem_pos_indexs = torch.arange(0, n, dtype=torch.long, device=device)
        self.kv_mem_pos_indexs_int32 = self.kv_mem_pos_indexs.to(torch.int32)
        self.kv_mem_use_state = torch.zeros(n, dtype=torch.int32, device=device)
        self._free_dev = torch.full((1,), n, dtype=torch.long, device=device)   # exact, on the device
        self._free_lb = n            # host lower bound of the free-row count
        self._free_exact = True      # the bound IS the count
        self._decision = torch.zeros(2, dtype=torch.long, device=device)
        self._scratch = torch.empty(int(L.lib().ll_kv_alloc_scratch_bytes(n)), dtype=torch.uint8, device=device)
        self._bump_cursor = 0
        self._bump_is_exact = True
        self.init_kv_buffers(n, head_dim, num_kv_heads, num_layers, dtype, device)

    def init_kv_buffers(self, max_num_tokens, head_dim, num_kv_heads, num_layers, dtype, device="cuda") -> None:
        """One ``[max_num_tokens, 2 * kv_heads, head_dim]`` tensor per layer, K heads first (:197-216)."""
        self.gpu_kv_buffer = [torch.empty((max_num_tokens, 2 * num_kv_heads, head_dim), dtype=dtype, device=device)
                              for _ in range(num_layers)]

    # ------------------------------------------------------------------ free-row count -- #
    @property
    def can_use_mem_size(self) -> int:
        """Rows currently free.  Exact; reads the device counter only if the host bound is stale."""
        if not self._free_exact:
            self._free_lb = int(self._free_dev.item())
            self._free_exact = True
        return self._free_lb

    def _fits(self, need: int) -> bool:
        """Admission check without a read-back whenever the lower bound already says yes."""
        if need <= self._free_lb:
            return True
        return need <= self.can_use_mem_size

    # ------------------------------------------------------------------ allocation ------ #
    def _device_alloc(self, need: int, mode: int) -> torch.Tensor:
        out = torch.empty(need, dtype=torch.int32, device=self.device)
        L.check(L.lib().ll_kv_alloc(self.kv_mem_use_state.data_ptr(), self.max_num_tokens, need, mode, out.data_ptr(),
                                    self._scratch.data_ptr(), self._decision.data_ptr(), self._free_dev.data_ptr(),
                                    L.stream_ptr()), "kv_alloc")
        return out

    @torch.no_grad()
    def alloc_kvcache(self, need_size):
        """``need_size`` free rows wherever they are (ascending); ``None`` if short (:219-231)."""
        if not self._fits(need_size):
            return None
        if need_size == 0:
            return self.kv_mem_pos_indexs[:0]
        out = self._device_alloc(need_size, 0)
        self._free_lb -= need_size
        return out.long()

    @torch.no_grad()
    def alloc_contiguous_kvcache(self, need_size):
        """``(select_index, start, end)`` of the first run of ``need_size`` consecutive free rows, or
        ``None`` (:234-267).  Reads the decision back (the return type carries python ints)."""
        if not self._fits(need_size) or need_size == 0:
            return None
        self._device_alloc(need_size, 2)
        mode, start = (int(v) for v in self._decision.tolist())
        if mode != 1:
            return None
        self._free_lb -= need_size
        return self.kv_mem_pos_indexs[start:start + need_size], start, start + need_size

    @torch.no_grad()
    def alloc_kvcache_index(self, need_size):
        """Rows for a prefill grid or a decode step: the append-only cursor while nothing was
        partially freed (no device reads, :286-292), else first contiguous run, else scattered --
        decided and filled on the device, no read-back.  int32 rows; ``None`` if the pool is short."""
        if self._bump_is_exact and self._bump_cursor + need_size <= self.max_num_tokens:
            start = self._bump_cursor
            self.kv_mem_use_state[start:start + need_size] += 1
            self._free_dev -= need_size
            self._free_lb -= need_size
            self._bump_cursor += need_size
            return self.kv_mem_pos_indexs_int32[start:start + need_size]
        if not self._fits(need_size):
            return None
        if need_size == 0:
            return self.kv_mem_pos_indexs_int32[:0]
        out = self._device_alloc(need_size, 1)
        self._free_lb -= need_size
        return out

    # ------------------------------------------------------------------ reference counts - #
    def _ref(self, token_index: torch.Tensor, delta: int) -> None:
        token_index = token_index.contiguous()
        L.require_cuda(token_index)
        L.check(L.lib().ll_kv_ref_update(self.kv_mem_use_state.data_ptr(), self.max_num_tokens, token_index.data_ptr(),
                                         token_index.numel(), L.index_width(token_index), delta,
                                         self._free_dev.data_ptr(), L.stream_ptr()), "kv_ref_update")

    @torch.no_grad()
    def add_ref(self, token_index: torch.Tensor):
        self._ref(token_index, +1)
        self._free_lb = max(self._free_lb - token_index.numel(), 0)   # at most this many rows left zero
        self._free_exact = False

    @torch.no_grad()
    def release_ref(self, token_index: torch.Tensor):
        self._bump_is_exact = False      # holes: the cursor no longer describes the free list
        self._ref(token_index, -1)
        self._free_exact = False         # the bound stays valid (releasing only frees rows)

    @torch.no_grad()
    def claim(self, num_rows: int) -> None:
        """Hand the first ``num_rows`` rows to an external owner (the slot layout, :336-353)."""
        if num_rows > self.can_use_mem_size:
            raise ValueError(f"cannot claim {num_rows} rows: only {self.can_use_mem_size} are free")
        self.kv_mem_use_state[:num_rows] += 1
        self._free_dev -= num_rows
        self._free_lb -= num_rows
        self._bump_cursor = max(self._bump_cursor, num_rows)

    @torch.no_grad()
    def free(self, free_index):
        self.release_ref(free_index.long())

    @torch.no_grad()
    def free_all(self):
        self.kv_mem_use_state.zero_()
        self._free_dev.fill_(self.max_num_tokens)
        self._free_lb, self._free_exact = self.max_num_tokens, True
        self._bump_cursor, self._bump_is_exact = 0, True
