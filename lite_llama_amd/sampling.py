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
    L.check(
        L.lib().ll_argmax(out.data_ptr(), flat.data_ptr(), flat.shape[0], v, flat.stride(0),
                          L.dtype_code(logits.dtype), L.stream_ptr()),
        "greedy_argmax",
    )
    return out.view(logits.shape[:-1])
