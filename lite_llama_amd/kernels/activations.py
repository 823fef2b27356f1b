"""relu / leaky_relu / tanh / gelu -- the four element-wise names the reference's kernel package exports
(lite_llama/kernels/activations.py:19-57: ``@triton.jit`` device helpers, no caller in its models).  Here: one HIP
kernel behind the C ABI (``ll_activation``, csrc/norm_act.hip) -- fp32 arithmetic, one rounding to the storage dtype; like
every other entry of this package there is no eager / CPU path (CPU tensors raise)."""

from __future__ import annotations

import torch

from .. import _lib as L


def _apply(x: torch.Tensor, kind: int, what: str) -> torch.Tensor:
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"{what}: fp16 / bf16 tensors only, got {x.dtype}")
    L.require_cuda(x)
    xc = x.contiguous()
    y = torch.empty_like(xc)
    L.check(L.lib().ll_activation(y.data_ptr(), xc.data_ptr(), xc.numel(), kind, L.dtype_code(x.dtype), L.stream_ptr()), what)
    return y.view(x.shape)


@torch.no_grad()
def relu(x):
    return _apply(x, 0, "relu")


@torch.no_grad()
def leaky_relu(x):
    return _apply(x, 1, "leaky_relu")


@torch.no_grad()
def tanh(x):
    return _apply(x, 2, "tanh")


@torch.no_grad()
def gelu(x):
    return _apply(x, 3, "gelu")
