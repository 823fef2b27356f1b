"""Quant-method strategies -- mirror of lite_llama/models/quantization/methods/*.py
(``LinearQuantMethod`` / ``MoeQuantMethod`` contract, base.py:23-74; registries
``_LINEAR_METHODS`` / ``_MOE_METHODS``, __init__.py:21-63).  ``apply`` dispatches the
layer's parameters to the HIP kernels; parameter names/dtypes are the reference's
(``weight`` + ``weight_scale``/``weight_zeros`` | ``weight_scale_inv``; experts
``gate_up_proj``/``down_proj`` + ``*_scale_inv``)."""

from __future__ import annotations

import os
from abc import ABC, abstractmethod

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..kernels import fused_moe, smoothquant_matmul, w4a16_matmul, w8a16_matmul
from ..kernels.quantization import smoothquant_gate_up_swiglu, smoothquant_matmul_partials, smoothquant_rows_matmul
from ..kernels.quantization import (dense16_linear, dense16_rows_linear, dense16_rows_wins, dense_matmul_partials, pack_w4a16_scales, pack_w4a16_weights, unpack_w4a16_weights, w4a16_matmul_partials,
                                    w4a16_matmul_prepacked_rows, w4a16_mtiled_supported,
                                    w4a16_matmul_prepacked, w4a16_prepacked_supported)
from .config import FP8, INT4, INT8, SMOOTHQUANT, QuantConfig
from .params import (
    quantize_fp8_per_channel,
    quantize_int4_groupwise,
    quantize_int8_groupwise,
    quantize_int8_per_channel,
)


def RawParameter(t: torch.Tensor) -> nn.Parameter:
    """Non-trainable parameter holding raw (integer / scale) storage (parameter.py:16-20)."""
    return nn.Parameter(t, requires_grad=False)


class LinearQuantMethod(ABC):
    @abstractmethod
    def create_weights(self, layer: nn.Module, input_size: int, output_size: int) -> None: ...

    @abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor) -> torch.Tensor: ...

    def convert_from_fp16(self, layer: nn.Module, quant: QuantConfig) -> None:
        raise NotImplementedError(f"{type(self).__name__} cannot be computed from fp16 weights at load time")


class MoeQuantMethod(ABC):
    @abstractmethod
    def create_weights(self, block: nn.Module) -> dict: ...

    @abstractmethod
    def apply(self, block, x, topk_weights, topk_ids, slots_ok: bool = False, aligned=None) -> torch.Tensor: ...

    def convert_from_fp16(self, block: nn.Module, quant: QuantConfig) -> None:
        raise NotImplementedError(f"{type(self).__name__} cannot be computed from fp16 weights at load time")


class UnquantizedLinearMethod(LinearQuantMethod):
    """fp16 weight through the library GEMM (hipBLASLt via ``F.linear``), unquantized.py:21-22."""

    def create_weights(self, layer, input_size, output_size):
        layer.weight = nn.Parameter(torch.empty(output_size, input_size, dtype=torch.float16), requires_grad=False)

    def apply(self, layer, x):
        m = x.numel() // x.shape[-1] if x.shape[-1] else 0
        if dense16_rows_wins(m, layer.weight.shape[0], layer.weight.shape[1]):  # wide outputs: the row-group loop (round 5)
            out = dense16_rows_linear(x, layer.weight, layer.bias)
            if out is not None:
                return out
        out = dense16_linear(x, layer.weight, layer.bias, policy="auto")  # decode shapes where the own kernel measured faster
        return out if out is not None else F.linear(x, layer.weight, layer.bias)

    def apply_gate_up_swiglu(self, layer, x):
        """``layer`` holds gate / up row-interleaved (linear.py::MergedColumnLinear): projection + ``silu(gate) * up`` as ONE launch
        of the 16-bit row-group kernel for decode shapes; ``None`` -> the caller runs the merged GEMM + ``swiglu_forward`` (which
        reads the interleaved pairs in place)."""
        m = x.numel() // x.shape[-1] if x.shape[-1] else 0
        if layer.bias is not None or not dense16_rows_wins(m, layer.weight.shape[0], layer.weight.shape[1]):
            return None
        return dense16_rows_linear(x, layer.weight, gate_up_swiglu=True)

    def apply_partials(self, layer, x, allow_bias: bool = False, max_splits: int = 12):
        """Decode-shaped projection left as fp32 split-K partials for its consumer (round 4, as the int4 method's): ``None``
        -> the caller runs :meth:`apply`."""
        if layer.bias is not None and not allow_bias:
            return None
        return dense_matmul_partials(x, layer.weight, max_splits=max_splits)

    def apply_finished(self, layer, x):
        """The measured policy for launches that write finished outputs (``dense16_wins``): own kernel where it beat the
        library GEMM, ``F.linear`` elsewhere."""
        return self.apply(layer, x)


def _ordered_input(layer, x):
    """Activation-ordered GPTQ linears (``layer.act_perm``, set by the checkpoint loader) keep their weight columns in
    group order; the activations are read through the same permutation (a gather of ``[tokens, K]``)."""
    perm = getattr(layer, "act_perm", None)
    return x if perm is None else x.index_select(-1, perm)


class W4A16LinearMethod(LinearQuantMethod):
    """methods/w4a16.py:18-43."""

    def create_weights(self, layer, input_size, output_size):
        q = layer.quant
        layer.weight = RawParameter(torch.empty(output_size, (input_size + 7) // 8, dtype=torch.int32))
        layer.weight_scale = RawParameter(torch.empty(*q.scale_shape(output_size, input_size), dtype=torch.float32))
        layer.weight_zeros = RawParameter(torch.empty(*q.scale_shape(output_size, input_size), dtype=torch.float32))

    def apply(self, layer, x):
        x = _ordered_input(layer, x)
        pre = self._prepacked(layer, x)
        if pre is not None:
            return w4a16_matmul_prepacked(x, pre, self._packed(layer), group_size=layer.quant.group_k, bias=layer.bias)
        rows = self._rows_layout(layer, x)
        if rows is not None:  # prefill-shaped call: the M-tiled engine over the same load-time layout (round 5)
            out = w4a16_matmul_prepacked_rows(x, rows, self._packed(layer), group_size=layer.quant.group_k, bias=layer.bias)
            if out is not None:
                return out
        return w4a16_matmul(x, self.reference_weight(layer), layer.weight_scale, layer.weight_zeros,
                            group_size=layer.quant.group_k, bias=layer.bias)

    def _rows_layout(self, layer, x):
        """The load-time layout for a call of MORE than 64 rows (``w4a16_matmul_prepacked_rows``): the compacted parameter
        itself, the copy the decode engine already cached, or -- outside a graph capture -- a freshly packed one (kept, like the
        decode engine's).  ``None`` -> the reference-layout engine (``LL_W4_NO_MTILED``, shapes off the 256-row grid, ...)."""
        m = x.numel() // x.shape[-1] if x.shape[-1] else 0
        if m <= 64 or os.environ.get("LL_W4_NO_MTILED") or not layer.weight.is_cuda:
            return None
        if getattr(layer, "_w4_compact_member", None) is not None:
            return None
        n, kp = layer.weight.shape
        if not w4a16_mtiled_supported(m, n, kp * 8, layer.quant.group_k) or self._packed(layer) is None:
            return None
        if getattr(layer, "_w4_compact", False):
            return self._as_packed(layer.weight.data)
        key = (layer.weight.data_ptr(), layer.weight._version)
        cached = getattr(layer, "_w4_prepacked", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if torch.cuda.is_current_stream_capturing() or os.environ.get("LL_W4_NO_PREPACK"):
            return None
        pre = pack_w4a16_weights(layer.weight.data)
        layer._w4_prepacked = (key, pre)
        return pre

    # ---- one resident copy of the int4 weights (round 4) ------------------------------------------------------------
    @staticmethod
    def _as_packed(t):
        n, kp = t.shape
        return t.view(n // 128, kp // 16, 8, 64, 4)

    def reference_weight(self, layer):
        """``qweight [N, K/8]`` in the reference layout: the parameter itself -- or, for a compacted layer, a transient
        tensor rebuilt from the load-time layout (bit-exact inverse permutation, ``ll_w4a16_unpack_weights``)."""
        member = getattr(layer, "_w4_compact_member", None)
        if member is not None:  # a member of a compacted merged storage: its rows of the merged reference tensor
            holder, rows = member
            return self.reference_weight(holder)[rows].contiguous()
        if getattr(layer, "_w4_compact", False):
            return unpack_w4a16_weights(self._as_packed(layer.weight.data))
        return layer.weight

    def compact(self, layer) -> int:
        """Make the decode engine's load-time layout the ONLY resident copy of the layer's int4 weights: ``layer.weight``
        keeps its shape and dtype but now ALIASES the packed storage (a permutation of the same words; graphs captured over
        the packed tensor stay valid), the reference-format tensor is freed and rebuilt on demand (:meth:`reference_weight`)
        for calls of more than 64 rows and for export.  Returns the bytes released (0: not served / already compact).
        ``state_dict()`` of a compacted model holds permuted words: call ``expand`` (or ``model.expand_weights()``) before
        saving or loading a checkpoint."""
        if getattr(layer, "_w4_compact", False) or getattr(layer, "_w4_compact_member", None) is not None:
            return 0
        w = layer.weight
        n, kp = w.shape
        if (not w.is_cuda or os.environ.get("LL_W4_NO_PREPACK") or torch.cuda.is_current_stream_capturing()
                or not w4a16_prepacked_supported(1, n, kp * 8, layer.quant.group_k) or self._packed(layer) is None):
            return 0
        cached = getattr(layer, "_w4_prepacked", None)
        pre = cached[1] if cached is not None and cached[0] == (w.data_ptr(), w._version) else pack_w4a16_weights(w.data)
        flat = pre.view(n, kp)
        if isinstance(w, nn.Parameter):
            w.data = flat
        else:
            layer.weight = flat
        layer._w4_compact = True
        layer._w4_prepacked = ((layer.weight.data_ptr(), layer.weight._version), pre)
        return n * kp * 4

    def expand(self, layer) -> None:
        """Undo :meth:`compact`: ``layer.weight`` is the reference-format tensor again (a fresh allocation)."""
        if not getattr(layer, "_w4_compact", False):
            return
        ref = unpack_w4a16_weights(self._as_packed(layer.weight.data))
        if isinstance(layer.weight, nn.Parameter):
            layer.weight.data = ref
        else:
            layer.weight = ref
        layer._w4_compact = False
        if hasattr(layer, "_w4_prepacked"):
            delattr(layer, "_w4_prepacked")

    def apply_partials(self, layer, x, allow_bias: bool = False, max_splits: int = 12):
        """Decode-shaped projection left as fp32 split-K partials for ``skip_rmsnorm_partials`` /
        ``decode_attention_partials`` (extension); ``None`` -> the caller runs :meth:`apply`.  The partials never
        include the bias: a consumer that adds it itself passes ``allow_bias``."""
        if (layer.bias is not None and not allow_bias) or os.environ.get("LL_W4_NO_PARTIALS"):
            return None
        x = _ordered_input(layer, x)
        pre = self._prepacked(layer, x)
        if pre is None:
            return None
        out = w4a16_matmul_partials(x, pre, self._packed(layer), group_size=layer.quant.group_k)
        return out if out is None or out.parts.shape[0] <= max_splits else None  # (the host plan never exceeds 12; q|k|v: 5)

    def apply_gate_up_swiglu(self, layer, x):
        """``layer`` holds gate/up row-interleaved (linear.py::MergedColumnLinear): one launch for
        both projections and the activation; ``None`` -> the caller falls back to the two-step form."""
        if layer.bias is not None:
            return None
        x = _ordered_input(layer, x)
        pre = self._prepacked(layer, x)
        if pre is not None:
            return w4a16_matmul_prepacked(x, pre, self._packed(layer), group_size=layer.quant.group_k,
                                          gate_up_swiglu=True)
        rows = self._rows_layout(layer, x)
        if rows is not None:  # prefill: the fused epilogue of the M-tiled engine
            return w4a16_matmul_prepacked_rows(x, rows, self._packed(layer), group_size=layer.quant.group_k, gate_up_swiglu=True)
        return None

    @staticmethod
    def _packed(layer):
        """Load-time re-layout of the scale/zero grids for the decode GEMM (extension; cached on the
        layer, rebuilt if the parameters were replaced).  Not built inside a graph capture."""
        key = (layer.weight_scale.data_ptr(), layer.weight_zeros.data_ptr(),
               layer.weight_scale._version, layer.weight_zeros._version)
        cached = getattr(layer, "_w4_packed", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if not layer.weight_scale.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        packed = pack_w4a16_scales(layer.weight_scale.data, layer.weight_zeros.data)
        layer._w4_packed = (key, packed)
        return packed

    def _prepacked(self, layer, x):
        """Decode-shaped calls (<= 64 rows) stream the weights from their load-time layout
        (``pack_w4a16_weights``; cached on the layer next to the reference-format parameter, which stays
        the checkpoint-facing tensor and serves every other shape).  ``None`` -> reference-format path."""
        n, kp = layer.weight.shape
        m = x.numel() // x.shape[-1] if x.shape[-1] else 0
        if getattr(layer, "_w4_compact_member", None) is not None:
            return None  # (a member of a compacted merged storage called on its own: reference route over rebuilt rows)
        if m < 1 or m > 64 or not layer.weight.is_cuda or (os.environ.get("LL_W4_NO_PREPACK") and not getattr(layer, "_w4_compact", False)):
            return None
        if not w4a16_prepacked_supported(m, n, kp * 8, layer.quant.group_k):
            return None
        if getattr(layer, "_w4_compact", False):  # the parameter IS the load-time layout
            return self._as_packed(layer.weight.data) if self._packed(layer) is not None else None
        key = (layer.weight.data_ptr(), layer.weight._version)
        cached = getattr(layer, "_w4_prepacked", None)
        if cached is not None and cached[0] == key:
            return cached[1] if self._packed(layer) is not None else None
        if torch.cuda.is_current_stream_capturing() or self._packed(layer) is None:
            return None
        pre = pack_w4a16_weights(layer.weight.data)
        layer._w4_prepacked = (key, pre)
        return pre

    def convert_from_fp16(self, layer, quant):
        qw, sc, zr = quantize_int4_groupwise(layer.weight.data, quant.group_k)
        layer.weight = RawParameter(qw)
        layer.weight_scale = RawParameter(sc)
        layer.weight_zeros = RawParameter(zr)


def _quantize_8bit(weight, quant: QuantConfig, in_size: int):
    if quant.format == FP8:
        return quantize_fp8_per_channel(weight)
    if quant.group_k < in_size:
        return quantize_int8_groupwise(weight, quant.group_k)
    return quantize_int8_per_channel(weight)


class W8A16LinearMethod(LinearQuantMethod):
    """methods/w8a16.py:39-63."""

    def create_weights(self, layer, input_size, output_size):
        q = layer.quant
        layer.weight = RawParameter(torch.empty(output_size, input_size, dtype=q.storage_dtype))
        layer.weight_scale_inv = RawParameter(torch.empty(*q.scale_shape(output_size, input_size), dtype=torch.float32))

    def apply(self, layer, x):
        q = layer.quant
        return w8a16_matmul(x, layer.weight, layer.weight_scale_inv, group_n=q.group_n,
                            group_k=min(q.group_k, layer.input_size), bias=layer.bias)

    def apply_partials(self, layer, x, allow_bias: bool = False, max_splits: int = 12):
        """Decode-shaped projection left as fp32 split-K partials (block scales applied) for its consumer; ``None`` -> the
        caller runs :meth:`apply`."""
        if layer.bias is not None and not allow_bias:
            return None
        q = layer.quant
        return dense_matmul_partials(x, layer.weight, layer.weight_scale_inv, group_n=q.group_n,
                                     group_k=min(q.group_k, layer.input_size), max_splits=max_splits)

    def convert_from_fp16(self, layer, quant):
        qw, sc = _quantize_8bit(layer.weight.data, quant, layer.input_size)
        layer.weight = RawParameter(qw)
        layer.weight_scale_inv = RawParameter(sc)


class SmoothQuantLinearMethod(LinearQuantMethod):
    """methods/w8a8.py:16-38 (dynamic per-token int8 activations x per-channel int8 weights)."""

    def create_weights(self, layer, input_size, output_size):
        q = layer.quant
        layer.weight = RawParameter(torch.empty(output_size, input_size, dtype=q.storage_dtype))
        layer.weight_scale_inv = RawParameter(torch.empty(*q.scale_shape(output_size, input_size), dtype=torch.float32))

    takes_int8_rows = True  # ``apply`` accepts kernels.norm_act.Int8Rows (the quantiser fused into the producing launch)

    def apply(self, layer, x):
        return smoothquant_matmul(x, layer.weight, layer.weight_scale_inv, bias=layer.bias)

    def apply_partials(self, layer, x, allow_bias: bool = False, max_splits: int = 12):
        """Decode-shaped projection left as exact int32 split-K planes + scales for ``skip_rmsnorm_q8`` (extension);
        ``None`` -> the caller runs :meth:`apply`."""
        if layer.bias is not None and not allow_bias:
            return None
        # (``allow_bias``: the consumer adds the bias itself -- the planes never carry it twice)
        return smoothquant_matmul_partials(x, layer.weight, layer.weight_scale_inv, bias=None if allow_bias else layer.bias,
                                           max_splits=max_splits)

    def apply_gate_up_swiglu(self, layer, x):
        """``layer`` holds gate/up row-interleaved: the int8 GEMM + ONE launch for the scale epilogue and the activation."""
        if layer.bias is not None:
            return None
        out = smoothquant_rows_matmul(x, layer.weight, layer.weight_scale_inv, gate_up_swiglu=True)  # wide outputs: one launch (round 5)
        return out if out is not None else smoothquant_gate_up_swiglu(x, layer.weight, layer.weight_scale_inv)

    def convert_from_fp16(self, layer, quant):
        qw, sc = quantize_int8_per_channel(layer.weight.data)
        layer.weight = RawParameter(qw)
        layer.weight_scale_inv = RawParameter(sc)


class UnquantizedMoeMethod(MoeQuantMethod):
    """methods/unquantized.py:25-57."""

    def create_weights(self, block):
        return {
            "gate_up_proj": nn.Parameter(torch.empty(block.num_experts, 2 * block.moe_intermediate_size,
                                                     block.hidden_size, dtype=torch.float16), requires_grad=False),
            "down_proj": nn.Parameter(torch.empty(block.num_experts, block.hidden_size,
                                                  block.moe_intermediate_size, dtype=torch.float16), requires_grad=False),
        }

    def apply(self, block, x, topk_weights, topk_ids, slots_ok: bool = False, aligned=None):
        return fused_moe(x, block.experts["gate_up_proj"], block.experts["down_proj"], topk_weights, topk_ids,
                         slots_ok=slots_ok, w1_interleaved=getattr(block, "_gu_interleaved", False), aligned=aligned)


class W8A16MoeMethod(MoeQuantMethod):
    """methods/w8a16.py:66-112."""

    @staticmethod
    def groups(block):
        """``((gn, gk) of gate|up, (gn, gk) of down)``: the config's blocks, or -- ``block.scale_cut`` (a TP shard that cuts
        them, model.py::SparseMoeBlock) -- the finer grid along the cut dimension (N of gate|up, K of down)."""
        q = block.quant
        cut = getattr(block, "scale_cut", 0)
        if cut:
            return (cut, q.group_k), (q.group_n, cut)
        return (q.group_n, q.group_k), (q.group_n, q.group_k)

    def create_weights(self, block):
        q = block.quant
        gu_n, gu_k = 2 * block.moe_intermediate_size, block.hidden_size
        d_n, d_k = block.hidden_size, block.moe_intermediate_size
        e = block.num_experts
        (g1n, g1k), (g2n, g2k) = self.groups(block)
        cdiv = lambda a, b: (a + b - 1) // b
        return {
            "gate_up_proj": RawParameter(torch.empty(e, gu_n, gu_k, dtype=q.storage_dtype)),
            "gate_up_proj_scale_inv": RawParameter(torch.empty(e, cdiv(gu_n, g1n), cdiv(gu_k, g1k), dtype=torch.float32)),
            "down_proj": RawParameter(torch.empty(e, d_n, d_k, dtype=q.storage_dtype)),
            "down_proj_scale_inv": RawParameter(torch.empty(e, cdiv(d_n, g2n), cdiv(d_k, g2k), dtype=torch.float32)),
        }

    def apply(self, block, x, topk_weights, topk_ids, slots_ok: bool = False, aligned=None):
        q = block.quant
        inter = getattr(block, "_gu_interleaved", False)  # rows paired (gate_j, up_j) at load time: scales are per row then
        if getattr(block, "scale_cut", 0) or (inter and getattr(block, "_gu_scale_rows", 1) > 1):
            g1, g2 = self.groups(block)
            g1n = 1 if inter else g1[0]
            return fused_moe(
                x, block.experts["gate_up_proj"], block.experts["down_proj"], topk_weights, topk_ids,
                w1_scale=block.experts["gate_up_proj_scale_inv"], w2_scale=block.experts["down_proj_scale_inv"],
                group_n=q.group_n, group_k=q.group_k, w1_group=(g1n, min(g1[1], block.hidden_size)),
                w2_group=(g2[0], min(g2[1], block.moe_intermediate_size)) if not getattr(block, "scale_cut", 0) else g2,
                slots_ok=slots_ok, w1_interleaved=inter, aligned=aligned,
            )
        return fused_moe(
            x, block.experts["gate_up_proj"], block.experts["down_proj"], topk_weights, topk_ids,
            w1_scale=block.experts["gate_up_proj_scale_inv"], w2_scale=block.experts["down_proj_scale_inv"],
            group_n=q.group_n, group_k=min(q.group_k, block.hidden_size), slots_ok=slots_ok, w1_interleaved=inter, aligned=aligned,
        )

    def convert_from_fp16(self, block, quant):
        for name in ("gate_up_proj", "down_proj"):
            qw, sc = _quantize_8bit(block.experts[name].data, quant, block.hidden_size)
            block.experts[name] = RawParameter(qw)
            block.experts[f"{name}_scale_inv"] = RawParameter(sc)


_LINEAR_METHODS = {FP8: W8A16LinearMethod, INT8: W8A16LinearMethod, SMOOTHQUANT: SmoothQuantLinearMethod,
                   INT4: W4A16LinearMethod}
# smoothquant experts are weight-only int8 (grouped GEMM keeps fp16 activations); int4 experts are
# rejected -- there is no grouped int4 GEMM (methods/__init__.py:28-36)
_MOE_METHODS = {FP8: W8A16MoeMethod, INT8: W8A16MoeMethod, SMOOTHQUANT: W8A16MoeMethod}


def get_linear_method(quant: QuantConfig | None) -> LinearQuantMethod:
    if quant is None:
        return UnquantizedLinearMethod()
    cls = _LINEAR_METHODS.get(quant.format)
    if cls is None:
        raise ValueError(f"no linear quant method for format {quant.format!r}")
    return cls()


def get_moe_method(quant: QuantConfig | None) -> MoeQuantMethod:
    if quant is None:
        return UnquantizedMoeMethod()
    cls = _MOE_METHODS.get(quant.format)
    if cls is None:
        raise ValueError(
            f"format {quant.format!r} is not supported for MoE experts; "
            "add the layer to modules_to_not_convert or use an 8-bit scheme"
        )
    return cls()
