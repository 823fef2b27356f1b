"""Ingestion of AutoAWQ / AutoGPTQ int4 checkpoint tensors into the path's W4A16 parameters
(SURVEY 8f-4).  The reference reaches W4A16 only by re-quantising fp16 weights; its loader maps
``qweight / qzeros / scales`` keys to unknown parameters (models/weights.py:166-173,266-268).  Here
the three tensors of one linear are converted on the device, bit-exactly, into what
``w4a16_matmul`` consumes (``weight`` int32 [N, K/8], ``weight_scale`` / ``weight_zeros`` fp32
[N, K/g] -- methods/w4a16.py:18-27), after which the layer runs the ordinary int4 path.

Formats (third-party, not part of the reference; restated from their published packers):
AutoAWQ 0.2.x GEMM and AutoGPTQ 0.7.x -- see include/lite_llama_amd.h.  Activation-ordered GPTQ tensors
(``desc_act``: ``g_idx[k]`` = group of input channel k, any order) are brought into group order at load time
(:func:`gptq_sort_groups`); the layer then reads its activations through the same permutation.
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
def gptq_sort_groups(qweight, g_idx, group_size: int = 128):
    """AutoGPTQ ``qweight [K/8, N]`` whose input channels belong to groups ``g_idx [K]`` ->
    ``(qweight', perm)`` with the channels in group order: ``perm = stable argsort(g_idx)``, row j of the unpacked
    ``qweight'`` is row ``perm[j]`` of the original, so ``g_idx[perm[j]] == j // group_size`` and
    ``sum_j x[perm[j]] W'[j] == sum_k x[k] W[k]``.  Returns ``(qweight, None)`` when the channels already are in
    group order.  Load-time only (plain tensor ops on the tensor's device)."""
    k = qweight.shape[0] * 8
    g_idx = g_idx.view(-1).to(torch.int64)
    if g_idx.numel() != k:
        raise ValueError(f"gptq_sort_groups: g_idx has {g_idx.numel()} entries for {k} input channels")
    want = torch.arange(k, device=g_idx.device) // group_size
    if torch.equal(g_idx, want):
        return qweight, None
    perm = torch.argsort(g_idx, stable=True)
    if not torch.equal(g_idx[perm], want):
        raise ValueError("gptq_sort_groups: g_idx does not hold exactly group_size channels per group")
    shifts = torch.arange(0, 32, 4, device=qweight.device, dtype=torch.int32).view(1, 8, 1)
    nib = ((qweight.unsqueeze(1) >> shifts) & 0xF).reshape(k, -1)            # [K, N]: row 8i + j = nibble j of word row i
    nib = nib.index_select(0, perm.to(qweight.device)).view(k // 8, 8, -1).to(torch.int64)
    packed = (nib << shifts.to(torch.int64)).sum(dim=1)                       # disjoint nibbles: the sum is the OR
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    return packed.contiguous(), perm.to(qweight.device)


@torch.no_grad()
def gptq_to_w4a16(qweight, qzeros, scales, g_idx=None, group_size: int = 128, checkpoint_format: str = "gptq"):
    """AutoGPTQ tensors (qweight [K/8, N], qzeros [K/g, N/8], scales fp16 [K/g, N]) -> the same triple.
    ``checkpoint_format`` "gptq" (v1: stored zero = z - 1) or "gptq_v2" (stored zero = z).  The device kernel reads
    channel k's group as ``k // group_size``: an activation-ordered ``g_idx`` raises here -- bring the tensor into
    group order first (:func:`gptq_sort_groups`; ``load_int4_checkpoint_linear`` and the checkpoint loader do)."""
    k, n = qweight.shape[0] * 8, qweight.shape[1]
    if checkpoint_format not in ("gptq", "gptq_v2"):
        raise ValueError(f"gptq_to_w4a16: unknown checkpoint_format {checkpoint_format!r}")
    if g_idx is not None:
        want = torch.arange(k, device=g_idx.device) // group_size
        if g_idx.numel() != k or not torch.equal(g_idx.to(want.dtype).view(-1), want):
            raise NotImplementedError("gptq_to_w4a16: activation-ordered tensor (desc_act): call gptq_sort_groups first")
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
        perm = None
        if g_idx is not None:
            qweight, perm = gptq_sort_groups(qweight, g_idx, group_size)
        w, s, z = gptq_to_w4a16(qweight, qzeros, scales, None, group_size, fmt)
        layer.act_perm = perm
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
