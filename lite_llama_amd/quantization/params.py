"""Weight quantisers that define the on-device formats the kernels read -- mirror of
lite_llama/models/quantization/params/{int4.py:12-49, int8.py:12-53, fp8.py:16-30}.
Plain torch tensor ops (they run wherever the weight lives, like the reference's)."""

from __future__ import annotations

import torch

FP8_E4M3_MAX = 448.0


def _div_const(t: torch.Tensor, c: float) -> torch.Tensor:
    """``t / c`` as a true IEEE division on every device.  torch's device kernels turn a division by a Python scalar into
    a multiplication by its reciprocal (1 ulp off for a few values), which made scales quantised on the GPU differ from the
    reference's CPU quantisers in the last bit and ~1 nibble in 1400 by one step (round 3, tools/diag_quant.py); dividing
    by a 0-dim tensor on the same device takes the tensor-tensor kernel and reproduces the CPU values bit for bit."""
    return t / torch.full((), c, dtype=t.dtype, device=t.device)


def quantize_int4_groupwise(weight: torch.Tensor, group_size: int = 128):
    """``[N, K]`` -> ``(qweight int32 [N, K/8] (8 nibbles per word, LSB first along K),
    scales fp32 [N, K/g], zeros fp32 [N, K/g])`` with ``scale = (max-min).clamp(1e-5)/14``,
    ``zero = round(-min/scale).clamp(0,15)``, ``q = round(w/scale + zero).clamp(0,15)``."""
    n, k = weight.shape
    if k % group_size != 0:
        raise ValueError(f"in_features {k} must be a multiple of group_size {group_size}")
    w = weight.float().reshape(n, k // group_size, group_size)
    lo = w.amin(dim=-1)
    hi = w.amax(dim=-1)
    scale = _div_const((hi - lo).clamp(min=1e-5), 14.0)
    zero = (-lo / scale).round().clamp(0, 15)
    q = (w / scale.unsqueeze(-1) + zero.unsqueeze(-1)).round().clamp(0, 15).to(torch.int32)
    q = q.reshape(n, -1, 8)
    packed = torch.zeros(n, k // 8, dtype=torch.int32, device=weight.device)
    for j in range(8):
        packed |= q[:, :, j] << (4 * j)  # int32 wrap-around of nibble 7 is the intended bit pattern
    return packed, scale.float(), zero.float()


def quantize_int8_per_channel(weight: torch.Tensor):
    scale = _div_const(weight.abs().amax(dim=-1, keepdim=True).float(), 127.0)
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    return (weight.float() / scale).round().clamp_(-127, 127).to(torch.int8), scale


def quantize_int8_groupwise(weight: torch.Tensor, group_size: int = 128):
    k = weight.shape[-1]
    if k % group_size != 0:
        raise ValueError(f"in_features {k} must be a multiple of group_size {group_size}")
    w = weight.float().unflatten(-1, (k // group_size, group_size))
    scale = _div_const(w.abs().amax(dim=-1), 127.0)
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = (w / scale.unsqueeze(-1)).round().clamp_(-127, 127).to(torch.int8)
    return q.flatten(-2), scale


def quantize_fp8_per_channel(weight: torch.Tensor):
    scale = _div_const(weight.abs().amax(dim=-1, keepdim=True).float(), FP8_E4M3_MAX)
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = (weight.float() / scale).clamp_(-FP8_E4M3_MAX, FP8_E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale
