"""Decode-step driver: per-step metadata, the token-attention KV pool and the hipGraph-captured
decode step.  Counterpart of lite_llama/executor/{attention_metadata.py:19-44,
kv_cache_manager.py:197-299, model_runner.py:153-218,310-330, cuda_graph.py:54-249} and of the
one-shot decode loop engine/llm_engine.py:137-213 (greedy, no EOS stop, lockstep batch).

MI355X-first differences from the reference (extensions, not parity changes):
  * the WHOLE step -- forward, greedy argmax, KV-row bump, ``b_seq_len += 1``,
    ``update_kv_index`` and feeding the sampled token back -- is captured in one hipGraph, so a
    decode step is a single ``hipGraphLaunch`` with no host work in between (the reference
    replays only the forward and does 5 host-issued copies + sampler + index update per step);
  * the graph is also captured under tensor parallelism (RCCL all-reduces are capturable);
    the reference disables graphs for TP>1 (model_runner.py:260-266).
"""

from __future__ import annotations

from dataclasses import dataclass, field

import os

import torch

from .. import _lib as L
from ..kernels import update_kv_index
from ..sampling import Sampler, greedy_argmax


@dataclass
class AttentionMetadata:
    """The struct every attention kernel call reads (same field names as the reference)."""

    kv_buffer: list = field(default_factory=list)        # per layer [max_tokens, 2*Hkv, D]
    cur_select_index: torch.Tensor | None = None          # rows written this step (int32)
    b_req_tokens_table: torch.Tensor | None = None        # [max_requests, max_seq_len] int32
    b_start_loc: torch.Tensor | None = None               # prefill only
    b_req_idx: torch.Tensor | None = None
    b_seq_len: torch.Tensor | None = None
    max_actual_seq_len: int = 0
    kv_scales: tuple = (1.0, 1.0)                          # fp8 KV pool only (extension): stored value * scale = K / V


@dataclass
class _Span:
    """The reference's GeneratedSpan (sampler.py:64-75): padded generated tokens + validity mask."""

    token_ids: torch.Tensor
    mask: torch.Tensor


class KVPool:
    """Token-granular KV pool: one ``[max_tokens, 2*Hkv, D]`` fp16 tensor per layer (K heads first),
    with the reference's contiguous bump allocator (kv_cache_manager.py:197-216,286-292)."""

    def __init__(self, num_layers: int, max_tokens: int, num_kv_heads: int, head_dim: int, device,
                 dtype=torch.float16):
        self.max_tokens = max_tokens
        self.device = device
        self.kv_buffer = [torch.zeros(max_tokens, 2 * num_kv_heads, head_dim, dtype=dtype, device=device)
                          for _ in range(num_layers)]
        self._next = 0

    def alloc(self, n: int) -> torch.Tensor:
        if self._next + n > self.max_tokens:
            raise RuntimeError(f"KV pool exhausted: need {n} rows, {self.max_tokens - self._next} free")
        idx = torch.arange(self._next, self._next + n, dtype=torch.int32, device=self.device)
        self._next += n
        return idx

    def reset(self) -> None:
        self._next = 0

    def claim(self, num_rows: int) -> None:
        """Hand the first ``num_rows`` rows to an external owner (the slot layout); the bump cursor
        resumes just past them (kv_cache_manager.py:336-353)."""
        if num_rows > self.max_tokens:
            raise ValueError(f"cannot claim {num_rows} rows: the pool holds {self.max_tokens}")
        self._next = max(self._next, num_rows)

    @property
    def used(self) -> int:
        return self._next


class DecodeEngine:
    """One-shot batch generation: padded-grid prefill, then lockstep greedy decode.

    ``prefill`` mirrors ModelRunner.prefill_alloc_kv_cache/_init_req_tokens_table
    (model_runner.py:153-198): the ``[batch, max_prompt_len]`` grid is flattened row-major,
    ``b_start_loc[i] = i * max_prompt_len``; ``decode`` mirrors decode_alloc_kv_cache (:200-218):
    after every forward the next B rows are bump-allocated, ``b_seq_len += 1`` and the table is
    updated, so at ``flash_decoding`` time ``b_seq_len`` already counts the token being fed.
    """

    def __init__(self, model, max_batch: int, max_seq_len: int, device="cuda", kv_dtype=torch.float16,
                 kv_block_size: int | None = None, kv_scales: tuple = (1.0, 1.0)):
        """``kv_block_size`` (extension): hand the KV pool out in blocks of that many rows from a device-side
        free stack (:class:`PagedKVPool`) instead of the reference's bump allocator; the kernels read the same per-token
        table either way, so the generated tokens are identical."""
        geo = model.geo
        self.model = model
        self.device = device
        self.max_batch = max_batch
        self.max_seq_len = max_seq_len
        at0 = model.layers[0].self_attn
        self.paged = kv_block_size is not None
        if self.paged:
            from .paged_kv import PagedKVPool

            per_req = (max_seq_len + kv_block_size - 1) // kv_block_size
            self.pool = PagedKVPool(geo.num_layers, max_batch * per_req + 1, kv_block_size, at0.num_kv_heads, geo.head_dim,
                                    device, max_batch, max_seq_len, kv_dtype)
        else:
            self.pool = KVPool(geo.num_layers, max_batch * max_seq_len, at0.num_kv_heads, geo.head_dim, device, kv_dtype)
        self.info = AttentionMetadata(kv_buffer=self.pool.kv_buffer, kv_scales=tuple(float(v) for v in kv_scales))
        self.info.b_req_tokens_table = torch.zeros(max_batch, max_seq_len, dtype=torch.int32, device=device)
        self._graph = None
        self._graph_key = None
        self._sampling = None
        if hasattr(model, "rotary_emb") and torch.device(device).type == "cuda":
            model.rotary_emb.ensure(max_seq_len + 1, device, model.embed_tokens.weight.dtype)

    # ------------------------------------------------------------------ prefill ----- #
    @torch.no_grad()
    def prefill(self, prompt_ids: torch.Tensor, prompt_lens: torch.Tensor | None = None, sampling=None) -> torch.Tensor:
        """``prompt_ids [B, Lp]`` (right-padded) -> next token per row ``[B]``: greedy, or -- with ``sampling`` -- drawn by
        the same sampler as every later token (llm_engine.py:168-176: the prefill step samples like a decode step; no
        token has been generated yet, so the repetition penalty has nothing to act on)."""
        b, lp = prompt_ids.shape
        dev = self.device
        self.pool.reset()
        if prompt_lens is None:
            prompt_lens = torch.full((b,), lp, dtype=torch.int32, device=dev)
        info = self.info
        info.b_req_idx = torch.arange(b, dtype=torch.int32, device=dev)
        info.b_seq_len = prompt_lens.to(torch.int32).clone()
        info.max_actual_seq_len = lp
        info.b_start_loc = torch.arange(b, dtype=torch.int32, device=dev) * lp
        if self.paged:
            # blocks for the valid tokens only; the grid's pad positions write their junk K/V into the junk block
            info.cur_select_index = self.pool.admit(info.b_req_idx, info.b_seq_len, lp, info.b_req_tokens_table)
        else:
            info.cur_select_index = self.pool.alloc(b * lp)
            # table[i, j] = row of token j of sequence i on the padded grid, valid tokens only
            # (_init_req_tokens_table, model_runner.py:153-179); a sequence's pad rows hold junk K/V
            # that attention never reads, and its own decode steps name fresh rows
            valid = torch.arange(lp, device=dev).unsqueeze(0) < info.b_seq_len.view(b, 1)
            info.b_req_tokens_table[:b, :lp] = torch.where(valid, info.cur_select_index.view(b, lp),
                                                           info.b_req_tokens_table[:b, :lp])
        position_ids = torch.arange(lp, device=dev).unsqueeze(0).expand(b, lp)
        rows = torch.arange(b, device=dev) * lp + (prompt_lens.long() - 1)
        last = self.model(prompt_ids, position_ids, info, logits_rows=rows)
        self._positions = prompt_lens.long().clone().view(b, 1)
        self._batch = b
        if sampling is None or sampling.temperature == 0.0:
            return greedy_argmax(last)
        uniform = torch.rand(b, device=dev)
        return Sampler().sample(last, sampling, None, uniform=uniform).view(-1)

    @torch.no_grad()
    def synthetic_context(self, batch: int, ctx_len: int, seed: int = 0, scattered: bool = False) -> torch.Tensor:
        """Benchmark helper: instead of running a prefill, fill the first ``ctx_len`` cache rows of
        every sequence with seeded random K/V (synthetic data of the prefill's shape) and return
        random first tokens.  Metadata ends up exactly as after ``prefill``.  ``scattered``: the
        table names the context rows in a random permutation (the fragmented-pool case: every K/V
        row of a sequence sits somewhere else) instead of one contiguous run per sequence."""
        dev = self.device
        self.pool.reset()
        info = self.info
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        info.b_req_idx = torch.arange(batch, dtype=torch.int32, device=dev)
        if self.paged:
            lens = torch.full((batch,), ctx_len, dtype=torch.int32, device=dev)
            rows = self.pool.admit(info.b_req_idx, lens, ctx_len, info.b_req_tokens_table).long()  # fills the table
            for kv in self.pool.kv_buffer:
                kv[rows] = (torch.randn(batch * ctx_len, kv.shape[1], kv.shape[2], generator=g, device=dev) * 0.5).to(kv.dtype)
        else:
            rows = self.pool.alloc(batch * ctx_len)
            if scattered:
                rows = rows[torch.randperm(batch * ctx_len, generator=g, device=dev)]
            info.b_req_tokens_table[:batch, :ctx_len] = rows.view(batch, ctx_len)
            for kv in self.pool.kv_buffer:
                kv[: batch * ctx_len].copy_(
                    (torch.randn(batch * ctx_len, kv.shape[1], kv.shape[2], generator=g, device=dev) * 0.5).to(kv.dtype))
        info.b_seq_len = torch.full((batch,), ctx_len, dtype=torch.int32, device=dev)
        info.max_actual_seq_len = ctx_len
        self._positions = torch.full((batch, 1), ctx_len, dtype=torch.long, device=dev)
        self._batch = batch
        return torch.randint(0, self.model.geo.vocab_size, (batch,), generator=g, device=dev)

    # ------------------------------------------------------------------ decode ------ #
    def _begin_decode(self, first_tokens: torch.Tensor, max_new_tokens: int):
        b, dev, info = self._batch, self.device, self.info
        # decode_alloc_kv_cache for the first decode step
        info.b_seq_len = info.b_seq_len + 1
        info.max_actual_seq_len += 1
        if self.paged:
            # the row of position seq_len - 1 (a block is popped on the device when the position opens one); every
            # later step repeats this launch inside the step (graph included) -- nothing is reserved ahead
            info.cur_select_index = self.pool.append(info.b_req_idx, info.b_seq_len, info.b_req_tokens_table)
        else:
            info.cur_select_index = self.pool.alloc(b)
            update_kv_index(info.b_req_tokens_table, info.b_req_idx, info.b_seq_len, info.cur_select_index)
            # reserve the rows of all remaining steps now: the bump allocator hands out the next B
            # rows each step, which the graph reproduces on device as ``cur_select_index += B``
            if max_new_tokens > 1:
                self.pool.alloc(b * (max_new_tokens - 1))
        self._input_ids = first_tokens.view(b, 1).clone()
        # column 0 holds the token sampled by the prefill step: it is part of the generated span the repetition
        # penalty looks at from the second token on (llm_engine.py:168-176: GeneratedSpan over tokens[:, :cur_pos])
        self._out = torch.zeros(b, max_new_tokens + 1, dtype=torch.long, device=dev)
        self._out[:, 0] = first_tokens.view(b)
        self._step = torch.ones(1, dtype=torch.long, device=dev)
        self._row_base = torch.arange(b, device=dev) * (max_new_tokens + 1)

    def _step_body(self):
        """forward -> greedy sample -> record -> advance metadata (all on device)."""
        info, b = self.info, self._batch
        logits = self.model(self._input_ids, self._positions, info)
        nxt = self._sample(logits[:, -1, :])
        self._advance(nxt)

    def _sample(self, logits: torch.Tensor) -> torch.Tensor:
        """Greedy argmax, or -- with ``sampling`` set by :meth:`decode` -- the reference sampler's sequence
        (repetition penalty over the tokens generated so far, then temperature + nucleus sampling) on
        the HIP kernels; the uniform numbers come from ``torch.rand`` (graph-safe Philox)."""
        sp = self._sampling
        if sp is None:
            return greedy_argmax(logits)
        b = self._batch
        generated = None
        if sp.repetition_penalty != 1.0:
            span = self._out.shape[1]
            mask = torch.arange(span, device=self.device).unsqueeze(0) < self._step  # [1, span] -> rows share it
            generated = _Span(self._out, mask.expand(b, span))
        uniform = torch.rand(b, device=self.device) if sp.temperature != 0.0 else None
        return Sampler().sample(logits, sp, generated, uniform=uniform).view(-1)

    def _advance(self, nxt: torch.Tensor) -> None:
        """record -> feed back -> positions/lengths/KV rows += -> token table: one launch when the state
        has the engine's own dtypes, else the reference-shaped sequence of tensor ops."""
        info, b = self.info, self._batch
        t = info.b_req_tokens_table
        if self.paged:
            self._out.view(-1).scatter_(0, self._row_base + self._step, nxt)
            self._step += 1
            self._input_ids.copy_(nxt.view(b, 1))
            self._positions += 1
            info.b_seq_len += 1
            self.pool.append(info.b_req_idx, info.b_seq_len, t, out=info.cur_select_index)
            return
        if (nxt.dtype == torch.int64 and info.cur_select_index.dtype == torch.int32 and info.b_seq_len.dtype == torch.int32
                and info.b_req_idx.dtype == torch.int32 and t.dtype == torch.int32 and nxt.is_contiguous()
                and info.cur_select_index.is_contiguous() and info.b_seq_len.is_contiguous()
                and info.b_req_idx.is_contiguous()):
            L.check(
                L.lib().ll_decode_advance(
                    self._out.data_ptr(), self._out.stride(0), self._step.data_ptr(), nxt.data_ptr(),
                    self._input_ids.data_ptr(), self._positions.data_ptr(), info.cur_select_index.data_ptr(),
                    info.b_seq_len.data_ptr(), info.b_req_idx.data_ptr(), t.data_ptr(), t.stride(0), t.stride(1),
                    b, L.stream_ptr()),
                "decode_advance")
            return
        self._out.view(-1).scatter_(0, self._row_base + self._step, nxt)
        self._step += 1
        self._input_ids.copy_(nxt.view(b, 1))
        self._positions += 1
        info.cur_select_index += b
        info.b_seq_len += 1
        update_kv_index(info.b_req_tokens_table, info.b_req_idx, info.b_seq_len, info.cur_select_index)

    @torch.no_grad()
    def decode(self, first_tokens: torch.Tensor, max_new_tokens: int, use_graph: bool = True,
               warmup_steps: int = 0, on_step=None, sampling=None, steps_per_graph: int = 1) -> torch.Tensor:
        """Generate ``max_new_tokens`` greedy tokens per row; returns ``[B, max_new_tokens]``.

        With ``use_graph`` the step is captured once (``max_actual_seq_len`` baked as the final
        context length rounded up to the attention partition size) and replayed per token.
        ``warmup_steps`` of the total are run before ``on_step`` timing hooks fire (bench use).
        ``steps_per_graph`` ("a whole multi-step decode is one graph launch", the counterpart of the loop in
        engine/llm_engine.py:137-213): that many consecutive steps are captured in ONE hipGraph -- every step already
        advances its own metadata on the device, so the steps chain without the host -- and a remainder shorter than the
        group replays a one-step graph; ``on_step(i)`` then fires once per replay with the index of its first step.
        """
        self._sampling = sampling  # None = greedy; else an object with temperature / top_p / repetition_penalty
        self._begin_decode(first_tokens, max_new_tokens)
        info = self.info
        final_len = info.max_actual_seq_len + max_new_tokens
        if final_len > self.max_seq_len:
            raise RuntimeError(f"context {final_len} exceeds max_seq_len {self.max_seq_len}")
        if use_graph:
            info.max_actual_seq_len = (final_len + 127) // 128 * 128  # bucket, fixes the attention grid
            # warm-up on a side stream (workspace growth, library init), fenced both ways
            # (cuda_graph.py:110-115); state is restored afterwards
            snap = self._snapshot()
            s = getattr(self, "_warm_stream", None)  # ONE warm-up stream per engine: eager scratch is kept per stream
            if s is None:
                s = self._warm_stream = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._step_body()
            torch.cuda.current_stream().wait_stream(s)
            self._restore(snap)
            # The sticky device-side error words are read here once too, right after the eager warm-up step created the
            # scratch they live in: a stale error of an earlier run surfaces before the run starts, and the check's first-use
            # costs in a process (first device-to-host copy: several ms) are paid outside anybody's timed region -- at the end
            # of a 32-step run they read as +0.2 ms per step (round 4).
            self._check_device_errors()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    self._step_body()
            except Exception:
                # a refused capture (e.g. a collective the backend cannot record) must not leave the stream in capture
                # mode: the caller falls back to eager launches on this very stream
                try:
                    if torch.cuda.is_current_stream_capturing():
                        graph.capture_end()
                except Exception:
                    pass
                self._restore(snap)
                raise
            self._restore(snap)  # capture does not execute, but keep the state explicit
            self._graph = graph
            group = max(1, min(int(steps_per_graph), max_new_tokens))
            multi = None
            if group > 1:
                multi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(multi):
                    for _ in range(group):
                        self._step_body()
                self._restore(snap)
                self._graph_multi = multi
            i = 0
            while i < max_new_tokens:
                if on_step is not None:
                    on_step(i)
                if multi is not None and max_new_tokens - i >= group:
                    multi.replay()
                    i += group
                else:
                    graph.replay()
                    i += 1
        else:
            for i in range(max_new_tokens):
                if on_step is not None:
                    on_step(i)
                info.max_actual_seq_len = int(info.max_actual_seq_len)  # eager: exact value
                self._step_body()
                info.max_actual_seq_len += 1
        self._check_device_errors()
        return self._out[:, 1:]

    def _check_device_errors(self) -> None:
        """Sticky device-side error words, read where the host synchronises anyway (end of a decode): a lost peer of the
        one-shot all-reduce, an exhausted / misused block pool.  Nothing to read (and no synchronisation) otherwise."""
        from ..distributed import parallel_state as ps

        if ps._ONESHOT is not None:
            ps.check_collective_errors()
        if L.gemm_scratch_error(torch.device(self.device)):
            raise RuntimeError("a split-K GEMM merge gave up waiting for a contributor workgroup: its output tile was poisoned "
                               "with NaN and the merge counters are not zero at rest")
        if self.paged:
            err = self.pool.error
            if err:
                raise RuntimeError(f"paged KV pool reported error bits {err:#x} (1: a call found too few free blocks, "
                                   "2: a request outgrew its block-table / token-table row)")

    def _snapshot(self):
        info = self.info
        return (self._input_ids.clone(), self._positions.clone(), info.cur_select_index.clone(),
                info.b_seq_len.clone(), self._step.clone(), self._out.clone(),
                info.b_req_tokens_table.clone(), self.pool.snapshot() if self.paged else None)

    def _restore(self, snap):
        info = self.info
        self._input_ids.copy_(snap[0])
        self._positions.copy_(snap[1])
        info.cur_select_index.copy_(snap[2])
        info.b_seq_len.copy_(snap[3])
        self._step.copy_(snap[4])
        self._out.copy_(snap[5])
        info.b_req_tokens_table.copy_(snap[6])
        if self.paged:
            self.pool.restore(snap[7])


from .slots import DEFAULT_BATCH_SIZES, DEFAULT_SEQ_LEN_BUCKETS, SlotBatch, SlotRunner, StepGraphs, slot_advance  # noqa: E402,F401
from .kv_cache_manager import KVCacheManager  # noqa: E402,F401
