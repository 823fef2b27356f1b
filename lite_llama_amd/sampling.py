"""Greedy sampling step after the hot path (reference engine/sampler.py:227-228,264:
``torch.argmax(logits, -1)``): first index of the row maximum, as a HIP kernel so the
decode step stays on one stream without a torch reduction in the graph."""

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
