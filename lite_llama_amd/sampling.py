"""Sampling step after the hot path -- mirror of lite_llama/engine/sampler.py: greedy argmax
(:227-228,264), HuggingFace-style repetition penalty (:77-115) and temperature + nucleus sampling
(:118-137), as HIP kernels so the decode step stays on one stream (the reference's nucleus path is
softmax + a full-vocabulary sort + cumsum + multinomial per step)."""

from __future__ import annotations

import torch

from . import _lib as L


def greedy_argmax(logits: torch.Tensor) -> torch.Tensor:
    """``[..., V] -> [...]`` int64 token ids (first maximum, like ``torch.argmax``)."""
    L.require_cuda(logits)
    v = logits.shape[-1]
    flat = logits.reshape(-1, v)
    if flat.stride(-1) != 1:
        flat = flat.contiguous()
    out = torch.empty(flat.shape[0], dtype=torch.int64, device=logits.device)
    rows = flat.shape[0]
    if v >= 16384 and rows < 1024:
        # vocabulary-sized rows at decode batch: spread each row over several workgroups
        chunks = max(1, min(64, 2048 // max(rows, 1), v // 4096))
        scratch = torch.empty(rows * chunks * 2, dtype=torch.int64, device=logits.device)
        L.check(
            L.lib().ll_argmax_split(out.data_ptr(), flat.data_ptr(), rows, v, flat.stride(0),
                                    L.dtype_code(logits.dtype), scratch.data_ptr(), chunks, L.stream_ptr()),
            "greedy_argmax",
        )
        return out.view(logits.shape[:-1])
    L.check(
        L.lib().ll_argmax(out.data_ptr(), flat.data_ptr(), flat.shape[0], v, flat.stride(0),
                          L.dtype_code(logits.dtype), L.stream_ptr()),
        "greedy_argmax",
    )
    return out.view(logits.shape[:-1])


def apply_repetition_penalty(logits: torch.Tensor, token_ids: torch.Tensor, mask: torch.Tensor, penalty) -> torch.Tensor:
    """``apply_repetition_penalty(logits, GeneratedSpan(token_ids, mask), penalty)`` of the reference:
    new ``[batch, vocab]`` logits, generated tokens divided by ``penalty`` when >= 0 and multiplied when
    negative (each token once).  ``penalty``: Python scalar (result keeps the logits dtype) or a
    ``[batch, 1]`` / ``[batch]`` float32 tensor (result float32 for fp16 logits -- torch's promotion)."""
    L.require_cuda(logits, token_ids, mask)
    if logits.dim() != 2 or token_ids.shape != mask.shape or token_ids.shape[0] != logits.shape[0]:
        raise ValueError("logits [batch, vocab], token_ids / mask [batch, span]")
    batch, vocab = logits.shape
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    ids = token_ids.to(torch.int64).contiguous()
    msk = mask.to(torch.bool).contiguous()
    if torch.is_tensor(penalty):
        L.require_cuda(penalty)
        pen_rows = penalty.reshape(-1).to(torch.float32).contiguous()
        if pen_rows.numel() != batch:
            raise ValueError("a tensor penalty needs one entry per row")
        out_dtype = torch.promote_types(logits.dtype, penalty.dtype)
        scalar = 1.0
    else:
        pen_rows, out_dtype, scalar = None, logits.dtype, float(penalty)
    out = torch.empty((batch, vocab), dtype=out_dtype, device=logits.device)
    L.check(
        L.lib().ll_repetition_penalty(
            out.data_ptr(), logits.data_ptr(), ids.data_ptr(), msk.data_ptr(), L.ptr(pen_rows), scalar, batch, vocab,
            ids.shape[1], logits.stride(0), out.stride(0), ids.stride(0), msk.stride(0),
            L.dtype_code(logits.dtype), L.dtype_code(out_dtype), L.stream_ptr()),
        "apply_repetition_penalty")
    return out


def sample_top_p(logits: torch.Tensor, temperature, top_p, uniform: torch.Tensor | None = None,
                 greedy: torch.Tensor | None = None) -> torch.Tensor:
    """Temperature + nucleus sampling of ``[batch, vocab]`` logits -> int64 ``[batch, 1]`` ids, the
    reference's ``sample_top_p(softmax(logits / temperature), top_p)`` in one launch and without the
    sort.  ``temperature`` / ``top_p``: scalars or per-row tensors; ``uniform``: one number in [0, 1)
    per row (default ``torch.rand``): the draw is the inverse CDF over the nucleus in token order;
    ``greedy`` (bool per row, optional): those rows return the first argmax instead."""
    L.require_cuda(logits, uniform, greedy)
    if logits.dim() != 2:
        raise ValueError("logits must be [batch, vocab]")
    batch, vocab = logits.shape
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    dev = logits.device

    def rows(v):
        if torch.is_tensor(v):
            t = v.reshape(-1).to(device=dev, dtype=torch.float32)
            return t.expand(batch).contiguous() if t.numel() == 1 else t.contiguous()
        return torch.full((batch,), float(v), dtype=torch.float32, device=dev)

    t, p = rows(temperature), rows(top_p)
    u = torch.rand(batch, device=dev) if uniform is None else uniform.reshape(-1).to(torch.float32).contiguous()
    g = None if greedy is None else greedy.reshape(-1).to(torch.bool).contiguous()
    if t.numel() != batch or p.numel() != batch or u.numel() != batch or (g is not None and g.numel() != batch):
        raise ValueError("per-row sampling knobs need one entry per row")
    out = torch.empty(batch, dtype=torch.int64, device=dev)
    L.check(
        L.lib().ll_sample_top_p(out.data_ptr(), logits.data_ptr(), t.data_ptr(), p.data_ptr(), u.data_ptr(), L.ptr(g),
                                batch, vocab, logits.stride(0), L.dtype_code(logits.dtype), L.stream_ptr()),
        "sample_top_p")
    return out.view(batch, 1)


class Sampler:
    """``Sampler.sample`` / ``sample_batched`` of the reference (sampler.py:199-270) on the kernels above.
    ``params``: any object with ``temperature``, ``top_p``, ``repetition_penalty`` (scalars for
    ``sample``; ``[batch, 1]`` tensors plus ``greedy`` / ``all_greedy`` / ``any_penalty`` for
    ``sample_batched``); ``generated``: object with ``token_ids`` and ``mask``."""

    @torch.no_grad()
    def sample(self, logits, params, generated=None, uniform=None):
        if logits.dim() == 3:
            logits = logits[:, -1, :]
        if params.repetition_penalty != 1.0 and generated is not None:
            logits = apply_repetition_penalty(logits, generated.token_ids, generated.mask, params.repetition_penalty)
        if params.temperature == 0.0:
            return greedy_argmax(logits).view(-1, 1)
        return sample_top_p(logits, params.temperature, params.top_p, uniform)

    @torch.no_grad()
    def sample_batched(self, logits, params, generated=None, uniform=None):
        if logits.dim() == 3:
            logits = logits[:, -1, :]
        if params.any_penalty and generated is not None:
            logits = apply_repetition_penalty(logits, generated.token_ids, generated.mask, params.repetition_penalty)
        if params.all_greedy:
            return greedy_argmax(logits).view(-1, 1)
        return sample_top_p(logits, params.temperature, params.top_p, uniform, greedy=params.greedy)
