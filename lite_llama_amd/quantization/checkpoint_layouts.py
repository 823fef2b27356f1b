"""Ingestion of AutoAWQ / AutoGPTQ int4 checkpoint tensors into the path's W4A16 parameters
(SURVEY 8f-4).  The reference reaches W4A16 only by re-quantising fp16 weights; its loader maps
``qweight / qzeros / scales`` keys to unknown parameters (models/weights.py:166-173,266-268).  Here
the three tensors of one linear are converted on the device, bit-exactly, into what
``w4a16_matmul`` consumes (``weight`` int32 [N, K/8], ``weight_scale`` / ``weight_zeros`` fp32
[N, K/g] -- methods/w4a16.py:18-27), after which the layer runs the ordinary int4 path.

Formats (third-party, not part of the reference; restated from their published packers):
AutoAWQ 0.2.x GEMM and AutoGPTQ 0.7.x without activation reordering -- see include/lite_llama_amd.h.
"""

from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L


def _check(qweight, qzeros, scales, k, n, group_size, what):
    L.require_cuda(qweight, qzeros, scales)
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        raise ValueError(f"{what}: qweight and qzeros must be int32, got {qweight.dtype} / {qzeros.dtype}")
    if scales.dtype != torch.float16:
        raise ValueError(f"{what}: scales must be float16, got {scales.dtype}")
    if k % 8 or n % 8 or k % group_size:
        raise ValueError(f"{what}: K={k} and N={n} must be multiples of 8 and K a multiple of group_size={group_size}")
    groups = k // group_size
    if tuple(qzeros.shape) != (groups, n // 8) or tuple(scales.shape) != (groups, n):
        raise ValueError(f"{what}: expected qzeros [{groups}, {n // 8}] and scales [{groups}, {n}], got "
                         f"{tuple(qzeros.shape)} and {tuple(scales.shape)}")
    dev = qweight.device
    return (torch.empty(n, k // 8, dtype=torch.int32, device=dev), torch.empty(n, groups, dtype=torch.float32, device=dev),
            torch.empty(n, groups, dtype=torch.float32, device=dev))


@torch.no_grad()
def awq_to_w4a16(qweight, qzeros, scales, group_size: int = 128):
    """AutoAWQ GEMM tensors (qweight [K, N/8], qzeros [K/g, N/8], scales fp16 [K/g, N]) ->
    ``(weight [N, K/8] int32, weight_scale [N, K/g] fp32, weight_zeros [N, K/g] fp32)``."""
    k, n = qweight.shape[0], qweight.shape[1] * 8
    out = _check(qweight, qzeros, scales, k, n, group_size, "awq_to_w4a16")
    qweight, qzeros, scales = qweight.contiguous(), qzeros.contiguous(), scales.contiguous()
    L.check(L.lib().ll_w4_from_awq(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), qweight.data_ptr(),
                                   qzeros.data_ptr(), scales.data_ptr(), k, n, group_size, L.stream_ptr()), "w4_from_awq")
    return out


@torch.no_grad()
def gptq_to_w4a16(qweight, qzeros, scales, g_idx=None, group_size: int = 128, checkpoint_format: str = "gptq"):
    """AutoGPTQ tensors (qweight [K/8, N], qzeros [K/g, N/8], scales fp16 [K/g, N]) -> the same triple.
    ``checkpoint_format`` "gptq" (v1: stored zero = z - 1) or "gptq_v2" (stored zero = z).  Activation
    reordering (a ``g_idx`` other than ``k // group_size``) is not supported and raises."""
    k, n = qweight.shape[0] * 8, qweight.shape[1]
    if checkpoint_format not in ("gptq", "gptq_v2"):
        raise ValueError(f"gptq_to_w4a16: unknown checkpoint_format {checkpoint_format!r}")
    if g_idx is not None:
        want = torch.arange(k, device=g_idx.device) // group_size
        if g_idx.numel() != k or not torch.equal(g_idx.to(want.dtype).view(-1), want):
            raise NotImplementedError("gptq_to_w4a16: activation-reordered checkpoints (desc_act) are not supported")
    out = _check(qweight, qzeros, scales, k, n, group_size, "gptq_to_w4a16")
    qweight, qzeros, scales = qweight.contiguous(), qzeros.contiguous(), scales.contiguous()
    L.check(L.lib().ll_w4_from_gptq(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), qweight.data_ptr(),
                                    qzeros.data_ptr(), scales.data_ptr(), k, n, group_size,
                                    1 if checkpoint_format == "gptq" else 0, L.stream_ptr()), "w4_from_gptq")
    return out


@torch.no_grad()
def load_int4_checkpoint_linear(layer, qweight, qzeros, scales, *, fmt: str, group_size: int = 128, g_idx=None,
                                bias=None):
    """Install one checkpoint linear into ``layer`` (a LinearBase): parameters become those of the
    int4 strategy and the layer's ``quant`` / ``quant_method`` are switched to it."""
    from . import QuantConfig, get_linear_method
    from .methods import RawParameter

    if fmt == "awq":
        w, s, z = awq_to_w4a16(qweight, qzeros, scales, group_size)
    elif fmt in ("gptq", "gptq_v2"):
        w, s, z = gptq_to_w4a16(qweight, qzeros, scales, g_idx, group_size, fmt)
    else:
        raise ValueError(f"unknown int4 checkpoint format {fmt!r} (awq, gptq, gptq_v2)")
    if tuple(w.shape) != (layer.output_size, layer.input_size // 8):
        raise ValueError(f"checkpoint linear is {w.shape[0]}x{w.shape[1] * 8}, layer expects "
                         f"{layer.output_size}x{layer.input_size}")
    quant = QuantConfig.int4_groupwise(group_size)
    layer.weight = RawParameter(w)
    layer.weight_scale = RawParameter(s)
    layer.weight_zeros = RawParameter(z)
    if bias is not None:
        layer.bias = nn.Parameter(bias.to(torch.float16), requires_grad=False)
    layer.quant, layer.quant_method = quant, get_linear_method(quant)
    return layer
