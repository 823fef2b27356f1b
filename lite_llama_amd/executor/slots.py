"""Continuous-batching side of the decode-step driver: fixed KV slot regions, per-step metadata and
the (batch size, context bucket) grid of captured decode steps.

Counterparts (behaviour, names and argument meaning kept so callers written against the reference
drive this unchanged):

  * :class:`SlotBatch`   -- lite_llama/executor/slot_batch.py:28-220
  * :class:`StepGraphs`  -- lite_llama/executor/cuda_graph.py:54-249 (CUDAGraphRunner + Manager)
  * :class:`SlotRunner`  -- the slice of lite_llama/executor/model_runner.py the slot path touches
    (``atten_info``, ``b_req_tokens_table``, ``graph_batch_size`` :240-248, ``enable_cuda_graph``
    :250-309, ``forward`` :310-330, ``enable_slot_kv_cache`` :225-238)

MI355X-first differences (extensions, not parity changes):
  * every captured step reads ONE set of persistent metadata vectors (views ``[:batch]`` of
    buffers sized for the largest batch), and :class:`SlotBatch` keeps its running-set metadata in
    those same vectors, so a steady-state replay copies nothing (the reference pushes five
    ``copy_`` per replay, cuda_graph.py:140-144);
  * the steady-state metadata advance is one launch (``ll_slot_advance``) instead of an add, a
    subtract and an indexed gather;
  * graphs are captured lazily on first use of a (batch, bucket) pair (``capture_all`` remains),
    and are also taken under tensor parallelism (RCCL all-reduces are capturable).
"""

from __future__ import annotations

from collections.abc import Sequence

import torch

from .. import _lib as L

# the reference's capture grid (cuda_graph.py:27-28)
DEFAULT_BATCH_SIZES: tuple[int, ...] = (1, 2, 4, 8, 16, 32, 64, 128)
DEFAULT_SEQ_LEN_BUCKETS: tuple[int, ...] = (256, 512, 1024, 2048, 4096)


def slot_advance(b_seq_len, b_req_idx, cur_select_index, table, positions=None, input_ids=None, next_tokens=None):
    """``b_seq_len += 1; cur_select_index = table[b_req_idx, b_seq_len - 1]`` (and optionally
    ``positions = b_seq_len - 1``, ``input_ids = next_tokens``) in one launch, in place."""
    L.require_cuda(b_seq_len, b_req_idx, cur_select_index, table)
    assert b_seq_len.dtype == b_req_idx.dtype and b_seq_len.dtype in (torch.int32, torch.int64)
    assert cur_select_index.dtype == torch.int32 and table.dtype == torch.int32
    for t in (b_seq_len, b_req_idx, cur_select_index):
        assert t.is_contiguous()
    n = b_seq_len.shape[0]
    assert b_req_idx.shape[0] == n and cur_select_index.shape[0] == n
    for t in (positions, input_ids, next_tokens):
        assert t is None or (t.dtype == torch.int64 and t.is_contiguous() and t.numel() == n)
    L.check(
        L.lib().ll_slot_advance(
            b_seq_len.data_ptr(), b_req_idx.data_ptr(), cur_select_index.data_ptr(),
            0 if positions is None else positions.data_ptr(), 0 if input_ids is None else input_ids.data_ptr(),
            0 if next_tokens is None else next_tokens.data_ptr(), table.data_ptr(), table.stride(0),
            table.stride(1), n, L.index_width(b_seq_len), L.stream_ptr()),
        "slot_advance")


class StepGraphs:
    """One captured decode step per ``(batch_size, seq_len_bucket)``; a graph fixes the input shapes
    and ``max_actual_seq_len`` (a host int that sizes the attention partitions), so both axes are
    enumerated (cuda_graph.py:22-28)."""

    def __init__(self, model, *, kv_buffer, b_req_tokens_table, batch_sizes=DEFAULT_BATCH_SIZES,
                 seq_len_buckets=DEFAULT_SEQ_LEN_BUCKETS, device="cuda"):
        from . import AttentionMetadata

        self.model = model
        self.kv_buffer = kv_buffer
        self.b_req_tokens_table = b_req_tokens_table
        self.batch_sizes = tuple(sorted(set(batch_sizes)))
        self.seq_len_buckets = tuple(sorted(set(seq_len_buckets)))
        self.device = device
        cap = self.batch_sizes[-1]
        # the persistent input surface shared by every graph (dtypes of cuda_graph.py:86-96)
        self.input_ids = torch.zeros(cap, 1, dtype=torch.long, device=device)
        self.position_ids = torch.zeros(cap, 1, dtype=torch.long, device=device)
        self.cur_select_index = torch.zeros(cap, dtype=torch.int32, device=device)
        self.b_seq_len = torch.zeros(cap, dtype=torch.long, device=device)
        self.b_req_idx = torch.zeros(cap, dtype=torch.long, device=device)
        self._meta = AttentionMetadata
        self._graphs: dict[tuple[int, int], tuple] = {}
        self._epoch = getattr(model, "layout_epoch", 0)  # the weight layouts the cached graphs were captured over

    # ------------------------------------------------------------------ selection --- #
    def _pick_bucket(self, current_max_seq_len: int) -> int | None:
        for bucket in self.seq_len_buckets:
            if bucket >= current_max_seq_len:
                return bucket
        return None

    def pad_to(self, batch_size: int) -> int | None:
        """Smallest captured batch size that can absorb ``batch_size`` (None: larger than the grid)."""
        for bs in self.batch_sizes:
            if bs >= batch_size:
                return bs
        return None

    # ------------------------------------------------------------------ capture ----- #
    def _info(self, bs: int, bucket: int):
        return self._meta(kv_buffer=self.kv_buffer, cur_select_index=self.cur_select_index[:bs],
                          b_req_tokens_table=self.b_req_tokens_table, b_start_loc=None,
                          b_req_idx=self.b_req_idx[:bs], b_seq_len=self.b_seq_len[:bs], max_actual_seq_len=bucket)

    def capture(self, bs: int, bucket: int):
        """Warm up on a side stream (fenced both ways), then record on the current stream.  The
        warm-up runs on lengths ``min(bucket, 32)``: at length 0 the attention would visit no K/V
        rows (and produce 0/0), cuda_graph.py:103-105.  The caller's metadata is put back after."""
        info = self._info(bs, bucket)
        ids, pos = self.input_ids[:bs], self.position_ids[:bs]
        saved = [t.clone() for t in (ids, pos, info.cur_select_index, info.b_seq_len, info.b_req_idx)]
        info.b_seq_len.fill_(min(bucket, 32))
        info.b_req_idx.copy_(torch.arange(bs, device=self.device) % self.b_req_tokens_table.shape[0])
        info.cur_select_index.copy_(self.b_req_tokens_table[info.b_req_idx, info.b_seq_len - 1])
        pos.fill_(min(bucket, 32) - 1)
        # the warm-up scatters junk K/V into the rows it names: keep and put back their contents
        rows = info.cur_select_index.long()
        kept = [kv[rows].clone() for kv in self.kv_buffer]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.model(ids, pos, info)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.model(ids, pos, info)
        for kv, k in zip(self.kv_buffer, kept):
            kv[rows] = k
        for dst, src in zip((ids, pos, info.cur_select_index, info.b_seq_len, info.b_req_idx), saved):
            dst.copy_(src)
        self._graphs[(bs, bucket)] = (graph, out)
        return graph, out

    def capture_all(self) -> None:
        for bs in self.batch_sizes:
            for bucket in self.seq_len_buckets:
                if (bs, bucket) not in self._graphs:
                    self.capture(bs, bucket)

    # ------------------------------------------------------------------ replay ------ #
    def try_replay(self, input_ids, position_ids, atten_info):
        """Run the matching captured step, or return None (prefill, batch off the grid, context past
        the largest bucket) so the caller runs eager (cuda_graph.py:218-249).  Inputs that already
        live in the persistent vectors are not copied."""
        batch_size, seq_len = input_ids.shape
        if seq_len != 1 or batch_size not in self.batch_sizes:
            return None
        bucket = self._pick_bucket(atten_info.max_actual_seq_len)
        if bucket is None:
            return None
        if getattr(self.model, "layout_epoch", 0) != self._epoch:
            # compact_weights / expand_weights ran since the capture: int4 storage was re-allocated or MoE rows re-ordered
            # under the recorded launches -- never replay those; the grid is re-captured on demand
            self._graphs.clear()
            self._epoch = getattr(self.model, "layout_epoch", 0)
        entry = self._graphs.get((batch_size, bucket))
        if entry is None:
            entry = self.capture(batch_size, bucket)
        graph, out = entry
        for dst, src in ((self.input_ids, input_ids), (self.position_ids, position_ids),
                         (self.cur_select_index, atten_info.cur_select_index),
                         (self.b_seq_len, atten_info.b_seq_len), (self.b_req_idx, atten_info.b_req_idx)):
            if src.data_ptr() != dst.data_ptr():
                dst[:batch_size].copy_(src.view(dst[:batch_size].shape))
        graph.replay()
        return out


class SlotRunner:
    """What the slot path needs of the reference's ModelRunner: the model, the token-attention KV
    pool, the request->rows table, the attention metadata struct and the captured-step grid."""

    def __init__(self, model, max_request_num: int, max_seq_len: int, device="cuda", kv_dtype=torch.float16):
        from . import AttentionMetadata, KVPool

        geo = model.geo
        self.model = model
        self.device = device
        self.max_seq_len = max_seq_len
        self.max_request_num = max_request_num
        at0 = model.layers[0].self_attn
        self.pool = KVPool(geo.num_layers, max_request_num * max_seq_len, at0.num_kv_heads, geo.head_dim, device,
                           kv_dtype)
        self.b_req_tokens_table = torch.zeros(max_request_num, max_seq_len, dtype=torch.int32, device=device)
        self.atten_info = AttentionMetadata(kv_buffer=self.pool.kv_buffer, b_req_tokens_table=self.b_req_tokens_table)
        self._graphs: StepGraphs | None = None
        self._slot_batch: SlotBatch | None = None
        if hasattr(model, "rotary_emb") and torch.device(device).type == "cuda":
            model.rotary_emb.ensure(max_seq_len + 1, device, model.embed_tokens.weight.dtype)

    def enable_graphs(self, batch_sizes=DEFAULT_BATCH_SIZES, seq_len_buckets=DEFAULT_SEQ_LEN_BUCKETS,
                      capture_all: bool = False) -> None:
        """Clamp the grid to what the table and ``max_seq_len`` can serve (model_runner.py:273-291:
        a batch larger than the table would index past it); a bucket ceiling above ``max_seq_len``
        is kept only as ``max_seq_len`` itself."""
        if self._graphs is not None:
            return
        buckets = tuple(b for b in seq_len_buckets if b <= self.max_seq_len)
        sizes = tuple(b for b in batch_sizes if b <= self.max_request_num)
        if not buckets or not sizes:
            return
        self._graphs = StepGraphs(self.model, kv_buffer=self.pool.kv_buffer, b_req_tokens_table=self.b_req_tokens_table,
                                  batch_sizes=sizes, seq_len_buckets=buckets, device=self.device)
        if capture_all:
            self._graphs.capture_all()

    enable_cuda_graph = enable_graphs  # the reference's name for the same switch

    def graph_batch_size(self, batch_size: int) -> int:
        if self._graphs is None:
            return batch_size
        return self._graphs.pad_to(batch_size) or batch_size

    def enable_slot_kv_cache(self) -> "SlotBatch":
        if self._slot_batch is None:
            self._slot_batch = SlotBatch(self)
        return self._slot_batch

    @torch.no_grad()
    def forward(self, input_ids, position_ids, multi_modal_inputs=None):
        if self._graphs is not None:
            out = self._graphs.try_replay(input_ids, position_ids, self.atten_info)
            if out is not None:
                return out
        return self.model(input_ids, position_ids, self.atten_info)


class SlotBatch:
    """Continuous-batching view of the KV pool: slot ``s`` permanently owns rows
    ``[s * max_seq_len, (s + 1) * max_seq_len)``, so the table is the identity map written once; the
    last slot backs the filler rows that pad a decode batch up to a captured size; ``b_req_idx`` /
    ``b_seq_len`` are rebuilt from the host only when the running set changes."""

    def __init__(self, runner) -> None:
        table = runner.b_req_tokens_table
        total_slots, row_len = (int(v) for v in table.shape)
        has_filler = total_slots > 1  # the table's last row backs the filler entries of a padded decode batch
        self._runner, self._atten = runner, runner.atten_info
        self.device, self.max_seq_len = runner.device, runner.max_seq_len
        self.num_slots = total_slots - 1 if has_filler else 1
        self._filler_slot: int | None = total_slots - 1 if has_filler else None
        table.copy_(torch.arange(total_slots * row_len, dtype=table.dtype, device=self.device).view(total_slots, row_len))
        if self._filler_slot is not None:
            start = self._filler_slot * row_len
            for layer in self._kv_layers():
                layer[start:start + row_len].zero_()
        self._claim(total_slots * row_len)
        self._row_offsets = torch.arange(total_slots, dtype=torch.int32, device=self.device)
        self._b_req_idx: torch.Tensor | None = None
        self._b_seq_len: torch.Tensor | None = None
        self._cur_select: torch.Tensor | None = None
        self._host_slots: list[int] = []
        self._host_lens: list[int] = []

    # the two runner shapes this drives: SlotRunner (pool) or a reference-shaped runner (kv_cache_manager)
    def _kv_layers(self):
        r = self._runner
        return r.pool.kv_buffer if hasattr(r, "pool") else r.kv_cache_manager.gpu_kv_buffer

    def _claim(self, rows: int) -> None:
        r = self._runner
        if hasattr(r, "pool"):
            r.pool.claim(rows)
        else:
            r.kv_cache_manager.claim(rows)

    # ------------------------------------------------------------------ steps ------- #
    def begin_prefill(self, slots: Sequence[int], prompt_lens: Sequence[int]) -> None:
        """Sequence ``i``'s token ``j`` of the row-major ``[n, max_prompt_len]`` grid lands in slot
        ``slots[i]``'s row ``j`` (pad positions write junk the sequence's own decode overwrites)."""
        longest = max(prompt_lens)
        if longest > self.max_seq_len:
            raise ValueError(f"prompt length {longest} exceeds max_seq_len {self.max_seq_len}")
        info = self._atten
        rows = self._to_device(slots)
        info.b_req_idx, info.b_seq_len = rows, self._to_device(prompt_lens)
        info.b_start_loc = self._row_offsets[: len(slots)] * longest           # packed grid: sequence i starts at i * longest
        info.cur_select_index = info.b_req_tokens_table[rows, :longest].reshape(-1)  # its scatter targets: the row's first columns
        info.max_actual_seq_len = longest
        self.reset()  # the next decode step rebuilds its vectors from the host

    def begin_decode(self, slots: Sequence[int], seq_lens: Sequence[int]) -> int:
        """``seq_lens``: length each sequence has AFTER this step's token (its K/V goes to row
        ``seq_lens[i] - 1``).  Returns the batch size actually submitted (>= ``len(slots)`` when
        padded up to a captured size; the caller drops the trailing logits rows)."""
        padded_slots, padded_lens = self._pad(slots, seq_lens)
        table = self._atten.b_req_tokens_table
        if padded_slots == self._host_slots and padded_lens == [n + 1 for n in self._host_lens]:
            # same requests, one token further along: advance on the device, nothing crosses PCIe
            _slot_advance(self._b_seq_len, self._b_req_idx, self._cur_select, table)
        else:
            n = len(padded_slots)
            self._b_req_idx, self._b_seq_len, self._cur_select = self._metadata_vectors(n)
            self._b_req_idx.copy_(torch.tensor(padded_slots, dtype=torch.long))
            self._b_seq_len.copy_(torch.tensor(padded_lens, dtype=torch.long))
            self._cur_select.copy_(table[self._b_req_idx, self._b_seq_len - 1])
        self._host_slots, self._host_lens = padded_slots, padded_lens
        self._atten.b_req_idx = self._b_req_idx
        self._atten.b_seq_len = self._b_seq_len
        self._atten.max_actual_seq_len = max(seq_lens)
        self._atten.cur_select_index = self._cur_select
        self._atten.b_start_loc = None
        return len(padded_slots)

    def reset(self) -> None:
        self._host_slots, self._host_lens = [], []

    @property
    def seq_lens(self) -> torch.Tensor:
        if self._b_seq_len is None:
            raise RuntimeError("begin_decode() must run before seq_lens is read")
        return self._b_seq_len

    # ------------------------------------------------------------------ internals --- #
    def _metadata_vectors(self, n: int):
        """(b_req_idx i64, b_seq_len i64, cur_select_index i32) of length ``n``: views of the captured
        steps' persistent vectors when the runner has them (a replay then copies nothing)."""
        g = getattr(self._runner, "_graphs", None)
        if g is not None and n <= g.b_req_idx.shape[0]:
            return g.b_req_idx[:n], g.b_seq_len[:n], g.cur_select_index[:n]
        return (torch.empty(n, dtype=torch.long, device=self.device), torch.empty(n, dtype=torch.long, device=self.device),
                torch.empty(n, dtype=torch.int32, device=self.device))

    def _pad(self, slots, seq_lens):
        slots, seq_lens = list(slots), list(seq_lens)
        if self._filler_slot is None:
            return slots, seq_lens
        pad = self._runner.graph_batch_size(len(slots)) - len(slots)
        if pad <= 0:
            return slots, seq_lens
        filler_len = min(max(seq_lens), self.max_seq_len)
        return slots + [self._filler_slot] * pad, seq_lens + [filler_len] * pad

    def _to_device(self, values) -> torch.Tensor:
        return torch.tensor(list(values), dtype=torch.long, device=self.device)


# indirection so the CPU tier can pin the host logic with the oracle's restatement of the kernel
_slot_advance = slot_advance
