"""Block-granular KV pool on the device (extension -- the reference's pool is token-granular with the
paging TODO at executor/kv_cache_manager.py:211).  Same storage as :class:`KVPool` (one ``[rows, 2*Hkv, D]`` tensor per
layer, rows = blocks x block_size), handed out in blocks; the per-token table the kernels read
(``b_req_tokens_table``) is derived on the device by ``ll_kv_paged_extend``, so the attention / KV-write kernels and
their results are those of token attention.  Nothing is read back: admission, the per-step append and release are
stream-ordered launches (the append sits inside the captured decode step); the host looks at ``free_blocks`` /
``error`` only when it chooses to synchronise."""

from __future__ import annotations

import torch

from .. import _lib as L


class PagedKVPool:
    def __init__(self, num_layers: int, num_blocks: int, block_size: int, num_kv_heads: int, head_dim: int, device,
                 max_requests: int, max_seq_len: int, dtype=torch.float16):
        if block_size < 1 or num_blocks < 2:
            raise ValueError("PagedKVPool needs block_size >= 1 and num_blocks >= 2 (one block is the junk block)")
        self.block_size, self.num_blocks, self.device = block_size, num_blocks, device
        self.max_requests, self.max_seq_len = max_requests, max_seq_len
        self.max_tokens = num_blocks * block_size
        self.kv_buffer = [torch.zeros(self.max_tokens, 2 * num_kv_heads, head_dim, dtype=dtype, device=device)
                          for _ in range(num_layers)]
        self.blocks_per_request = (max_seq_len + block_size - 1) // block_size
        self.free_stack = torch.empty(num_blocks - 1, dtype=torch.int32, device=device)
        self.state = torch.zeros(2, dtype=torch.int32, device=device)          # [free blocks, error flags]
        self.block_table = torch.zeros(max_requests, self.blocks_per_request, dtype=torch.int32, device=device)
        self.req_blocks = torch.zeros(max_requests, dtype=torch.int32, device=device)
        self.reset()

    def reset(self) -> None:
        L.check(L.lib().ll_kv_paged_reset(self.free_stack.data_ptr(), self.state.data_ptr(), self.req_blocks.data_ptr(),
                                          self.num_blocks, self.max_requests, L.stream_ptr()), "kv_paged_reset")

    def _extend(self, req_idx, lens, len_bias, grid_len, from_end, token_table, out=None):
        n = req_idx.numel()
        L.require_cuda(req_idx, lens, token_table)
        if req_idx.dtype != torch.int32 or lens.dtype != torch.int32 or token_table.dtype != torch.int32:
            raise ValueError("PagedKVPool: request indices, lengths and the token table must be int32")
        if token_table.stride(1) != 1 or not req_idx.is_contiguous() or not lens.is_contiguous():
            raise ValueError("PagedKVPool: the token table must be row-major, indices and lengths contiguous")
        select = out if out is not None else torch.empty(n * grid_len, dtype=torch.int32, device=self.device)
        if select.dtype != torch.int32 or select.numel() != n * grid_len or not select.is_contiguous():
            raise ValueError("PagedKVPool: the row output must be a contiguous int32 tensor of n * grid_len entries")
        L.check(L.lib().ll_kv_paged_extend(
            self.free_stack.data_ptr(), self.state.data_ptr(), self.block_table.data_ptr(), self.block_table.stride(0),
            self.req_blocks.data_ptr(), req_idx.data_ptr(), lens.data_ptr(), len_bias, n, self.block_size, grid_len,
            from_end, token_table.data_ptr(), token_table.stride(0), select.data_ptr(), self.num_blocks, L.stream_ptr()),
            "kv_paged_extend")
        return select

    @torch.no_grad()
    def admit(self, req_idx: torch.Tensor, lens: torch.Tensor, grid_len: int, token_table: torch.Tensor) -> torch.Tensor:
        """Blocks for ``lens[i]`` tokens of every listed request + the rows of the padded prefill grid ``[n, grid_len]``
        (``cur_select_index``; pad positions name junk rows); fills ``token_table[req, :len]``."""
        return self._extend(req_idx, lens, 0, grid_len, 0, token_table)

    @torch.no_grad()
    def append(self, req_idx: torch.Tensor, seq_len: torch.Tensor, token_table: torch.Tensor, out=None) -> torch.Tensor:
        """The decode step's row: position ``seq_len[i] - 1`` of every listed request (a new block is popped when the
        position opens one); writes ``token_table[req, seq_len - 1]`` and returns / fills ``cur_select_index``."""
        return self._extend(req_idx, seq_len, 0, 1, 1, token_table, out=out)

    @torch.no_grad()
    def release(self, req_idx: torch.Tensor) -> None:
        L.require_cuda(req_idx)
        L.check(L.lib().ll_kv_paged_release(self.free_stack.data_ptr(), self.state.data_ptr(), self.block_table.data_ptr(),
                                            self.block_table.stride(0), self.req_blocks.data_ptr(), req_idx.data_ptr(),
                                            req_idx.numel(), L.stream_ptr()), "kv_paged_release")

    # host-visible (synchronising) views
    @property
    def free_blocks(self) -> int:
        return int(self.state[0].item())

    @property
    def error(self) -> int:
        """0, or flags: 1 = a call found too few free blocks (it allocated nothing), 2 = a request outgrew max_seq_len."""
        return int(self.state[1].item())

    def snapshot(self):
        return self.state.clone(), self.req_blocks.clone(), self.block_table.clone()

    def restore(self, snap) -> None:
        """Undo appends made since ``snapshot`` (pops only move the stack top: the stack itself is intact)."""
        self.state.copy_(snap[0])
        self.req_blocks.copy_(snap[1])
        self.block_table.copy_(snap[2])
