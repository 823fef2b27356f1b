"""w4a16_matmul / w8a16_matmul / smoothquant_matmul -- mirror of
lite_llama/kernels/quantization/{w4a16.py:152-207, w8a16.py:155-216, w8a8.py:151-217}
over the HIP C-ABI.  Same signatures, same ValueErrors, same on-device weight formats
(the reference's own: qweight [N, K/8] int32 LSB-first nibbles + fp32 scales/zeros [N, K/g];
8-bit [N, K] uint8 (e4m3 bits) / int8 + fp32 scale grid)."""

from __future__ import annotations

import os

import torch

from .. import _lib as L


def _flatten(x: torch.Tensor, k: int):
    a = x.reshape(-1, k)
    if a.stride(-1) != 1 or (a.stride(0) % 8) != 0 or (a.data_ptr() % 16) != 0:
        a = a.contiguous()
    return a


def _auto_prepacked(m, n, k, group_size, qweight, scales, zeros):
    """The drop-in route's access to the decode engine (round 3): a caller of the REFERENCE signature (the reference's
    ``W4A16LinearMethod.apply`` after ``integration.install()``) passes the reference-format parameters on every call; for
    decode shapes (<= 64 rows) their load-time layouts -- ``pack_w4a16_weights`` / ``pack_w4a16_scales``, bit-exact
    permutations -- are built on first use and kept ON the weight tensor object (freed with it; rebuilt when any of the
    three tensors was replaced or written in place).  Costs a second copy of the int4 weights (+0.5625 B per weight);
    ``LL_W4_NO_AUTO_PREPACK=1`` keeps the reference-layout engine.  Never built inside a graph capture."""
    if m < 1 or m > 64 or os.environ.get("LL_W4_NO_AUTO_PREPACK") or not qweight.is_contiguous():
        return None
    if not w4a16_prepacked_supported(m, n, k, group_size):
        return None
    key = (qweight.data_ptr(), qweight._version, scales.data_ptr(), scales._version, zeros.data_ptr(), zeros._version)
    cached = getattr(qweight, "_ll_prepacked", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    if torch.cuda.is_current_stream_capturing():
        return None
    try:
        pre = (pack_w4a16_weights(qweight), pack_w4a16_scales(scales, zeros))
        qweight._ll_prepacked = (key, pre)
    except (AttributeError, RuntimeError):
        return None
    return pre


def w4a16_matmul(
    x: torch.Tensor,
    qweight: torch.Tensor,
    scales: torch.Tensor,
    zeros: torch.Tensor,
    *,
    group_size: int = 128,
    bias: torch.Tensor | None = None,
) -> torch.Tensor:
    """``x @ dequant(qweight).T (+ bias)``; ``W[n,k] = (nib(n,k) - zeros[n,k//g]) * scales[n,k//g]``.
    Decode-shaped calls (<= 64 rows) run on the pre-packed engine over load-time layouts built once per weight
    (:func:`_auto_prepacked`); every other shape on the generic engine over the reference layout."""
    if x.dtype != torch.float16:
        raise ValueError(f"w4a16 activations must be fp16, got {x.dtype}")
    if qweight.dtype != torch.int32:
        raise ValueError(f"qweight must be int32 (packed int4), got {qweight.dtype}")
    n, k_packed = qweight.shape
    k = k_packed * 8
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    if k % group_size != 0:
        raise ValueError(f"K ({k}) must be a multiple of group_size ({group_size})")
    L.require_cuda(x, qweight, scales, zeros, bias)
    leading = x.shape[:-1]
    a = _flatten(x, k)
    m = a.shape[0]
    if qweight.stride(1) != 1:
        qweight = qweight.contiguous()
    if scales.dtype != torch.float32 or scales.stride(1) != 1:
        scales = scales.float().contiguous()
    if zeros.dtype != torch.float32 or zeros.stride() != scales.stride():
        zeros = zeros.float().contiguous()
        scales = scales.contiguous()
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    pre = _auto_prepacked(m, n, k, group_size, qweight, scales, zeros)
    if pre is not None:  # decode-shaped call of the reference signature: the load-time layouts, made once per weight
        return w4a16_matmul_prepacked(x, pre[0], pre[1], group_size=group_size, bias=bias)
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    ws, cnt = L.gemm_workspace(x.device, m, n, k)
    L.check(
        L.lib().ll_w4a16_matmul(
            out.data_ptr(), a.data_ptr(), qweight.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
            L.ptr(bias), m, n, k, int(group_size), a.stride(0), qweight.stride(0), scales.stride(0),
            ws.data_ptr(), cnt.data_ptr(), L.stream_ptr(),
        ),
        "w4a16_matmul",
    )
    return out.reshape(*leading, n)


def pack_w4a16_scales(scales: torch.Tensor, zeros: torch.Tensor) -> torch.Tensor:
    """Load-time companion of :func:`w4a16_matmul`: ``[N, K/g]`` fp32 scales/zeros -> int32
    ``[K/g, N, 2]`` holding the fp16 pairs ``(s, s)``, ``(-z*s, -z*s)`` the GEMM feeds its dequant."""
    L.require_cuda(scales, zeros)
    if scales.shape != zeros.shape or scales.dim() != 2:
        raise ValueError("scales and zeros must both be [N, K/g]")
    scales = scales.float().contiguous()
    zeros = zeros.float().contiguous()
    n, groups = scales.shape
    packed = torch.empty((groups, n, 2), dtype=torch.int32, device=scales.device)
    L.check(L.lib().ll_w4a16_pack_scales(packed.data_ptr(), scales.data_ptr(), zeros.data_ptr(), n, groups,
                                         scales.stride(0), L.stream_ptr()), "pack_w4a16_scales")
    return packed


def pack_w4a16_weights(qweight: torch.Tensor) -> torch.Tensor:
    """Load-time companion of :func:`w4a16_matmul_prepacked` (the layout slot the reference reserves in
    ``models/quantization/_layout``): a bit-exact permutation of ``qweight [N, K/8]`` into the order the
    decode engine streams it -- per (128-row tile, 128-k chunk) one 8-KB block ``[wave 8][lane 64] x 16 B``,
    even nibbles of every word first.  ``N`` and ``K`` must be multiples of 128."""
    L.require_cuda(qweight)
    if qweight.dtype != torch.int32 or qweight.dim() != 2:
        raise ValueError("qweight must be int32 [N, K/8]")
    n, kp = qweight.shape
    k = kp * 8
    if n % 128 or k % 128:
        raise ValueError(f"pack_w4a16_weights needs N and K multiples of 128, got {n} x {k}")
    if qweight.stride(1) != 1 or qweight.stride(0) % 4 or qweight.data_ptr() % 16:
        qweight = qweight.contiguous()
    packed = torch.empty((n // 128, k // 128, 8, 64, 4), dtype=torch.int32, device=qweight.device)
    L.check(L.lib().ll_w4a16_pack_weights(packed.data_ptr(), qweight.data_ptr(), n, k, qweight.stride(0),
                                          L.stream_ptr()), "pack_w4a16_weights")
    return packed


def unpack_w4a16_weights(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_w4a16_weights` (bit-exact): the int32 ``[N, K/8]`` reference-format tensor."""
    L.require_cuda(packed)
    if packed.dtype != torch.int32 or packed.dim() != 5 or tuple(packed.shape[2:]) != (8, 64, 4) or not packed.is_contiguous():
        raise ValueError("packed must be the int32 [N/128, K/128, 8, 64, 4] tensor made by pack_w4a16_weights")
    n, k = packed.shape[0] * 128, packed.shape[1] * 128
    qweight = torch.empty((n, k // 8), dtype=torch.int32, device=packed.device)
    L.check(L.lib().ll_w4a16_unpack_weights(qweight.data_ptr(), packed.data_ptr(), n, k, qweight.stride(0), L.stream_ptr()),
            "unpack_w4a16_weights")
    return qweight


def w4a16_prepacked_supported(m: int, n: int, k: int, group_size: int) -> bool:
    return bool(L.lib().ll_w4a16_prepacked_supported(m, n, k, int(group_size)))


def _launch_prepacked(out_ptr, a, m, n, k, packed_weight, packed_scales, bias, group_size, epilogue, what):
    """The decode-engine launch."""
    ws, cnt = L.gemm_workspace(a.device, m, n, k)
    L.check(
        L.lib().ll_w4a16_matmul_prepacked(
            out_ptr, a.data_ptr(), packed_weight.data_ptr(), packed_scales.data_ptr(), L.ptr(bias), m, n, k,
            int(group_size), a.stride(0), ws.data_ptr(), cnt.data_ptr(), epilogue, L.stream_ptr(),
        ),
        what,
    )


def w4a16_matmul_prepacked(x, packed_weight, packed_scales, *, group_size: int = 128, bias=None, gate_up_swiglu=False,
                           _tile_blocks: int = 0):
    """Decode-engine form of :func:`w4a16_matmul` over the load-time layouts (``pack_w4a16_weights`` /
    ``pack_w4a16_scales``); at most 64 rows.  ``gate_up_swiglu`` applies the fused epilogue of
    the fused gate|up launch (rows interleaved gate/up: out = silu(gate) * up).  Same arithmetic as ``w4a16_matmul``.
    ``_tile_blocks`` (tests / tuning): 0 = tile width chosen by the host plan, 1 / 2 = 128- / 256-row tiles."""
    if x.dtype != torch.float16:
        raise ValueError(f"w4a16 activations must be fp16, got {x.dtype}")
    L.require_cuda(x, packed_weight, packed_scales, bias)
    if packed_weight.dtype != torch.int32 or packed_weight.dim() != 5 or not packed_weight.is_contiguous():
        raise ValueError("packed_weight must be the int32 [N/128, K/128, 8, 64, 4] tensor made by pack_w4a16_weights")
    n, k = packed_weight.shape[0] * 128, packed_weight.shape[1] * 128
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    if tuple(packed_scales.shape) != (k // group_size, n, 2) or packed_scales.dtype != torch.int32 \
            or not packed_scales.is_contiguous():
        raise ValueError("packed_scales must be the int32 [K/g, N, 2] tensor made by pack_w4a16_scales")
    leading = x.shape[:-1]
    a = _flatten(x, k)
    m = a.shape[0]
    if not w4a16_prepacked_supported(max(m, 1), n, k, group_size):
        raise ValueError(f"w4a16_matmul_prepacked: shape M={m} N={n} K={k} g={group_size} is outside the decode engine")
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    n_out = n // 2 if gate_up_swiglu else n
    out = torch.empty((m, n_out), dtype=x.dtype, device=x.device)
    _launch_prepacked(out.data_ptr(), a, m, n, k, packed_weight, packed_scales, bias, group_size,
                      (1 if gate_up_swiglu else 0) | ((int(_tile_blocks) & 3) << 8), "w4a16_matmul_prepacked")
    return out.reshape(*leading, n_out)


def w4a16_mtiled_supported(m: int, n: int, k: int, group_size: int) -> bool:
    return bool(L.lib().ll_w4a16_mtiled_supported(m, n, k, int(group_size)))


def w4a16_matmul_prepacked_rows(x, packed_weight, packed_scales, *, group_size: int = 128, bias=None, gate_up_swiglu=False):
    """:func:`w4a16_matmul` for MANY rows (prefill) over the decode engine's load-time layouts: an M-tiled MFMA GEMM that
    dequantises every weight tile once per 256 rows (csrc/gemm_w4_prefill.hip) -- the > 64-row calls of a compacted model no
    longer rebuild the reference-format tensor, and the generic engine's 64-row weight-streaming tile is not looped over M.
    Same arithmetic as the decode engines (bit-identical dequantisation, fp32 accumulation, fused bias / swiglu epilogues);
    ``None`` when the shape is not served (N not a multiple of 256, ...)."""
    if x.dtype != torch.float16:
        raise ValueError(f"w4a16 activations must be fp16, got {x.dtype}")
    L.require_cuda(x, packed_weight, packed_scales, bias)
    n, k = packed_weight.shape[0] * 128, packed_weight.shape[1] * 128
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    a = _flatten(x, k)
    m = a.shape[0]
    if m < 1 or not w4a16_mtiled_supported(m, n, k, group_size) or (gate_up_swiglu and n % 2):
        return None
    if (m - 1) * a.stride(0) * 2 + k * 2 >= 2 ** 31:
        return None
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    n_out = n // 2 if gate_up_swiglu else n
    out = torch.empty((m, n_out), dtype=x.dtype, device=x.device)
    L.check(L.lib().ll_w4a16_matmul_prepacked_mtiled(out.data_ptr(), a.data_ptr(), packed_weight.data_ptr(), packed_scales.data_ptr(),
                                                     L.ptr(bias), m, n, k, int(group_size), a.stride(0), 1 if gate_up_swiglu else 0,
                                                     L.stream_ptr()), "w4a16_matmul_prepacked_rows")
    return out.reshape(*x.shape[:-1], n_out)


def w4a16_matmul_partials(x, packed_weight, packed_scales, *, group_size: int = 128, _unit_loop_engine: bool = False):
    """Decode-step extension: the projection as ``S`` fp32 split-K partial sums (:class:`PartialSums`) for
    :func:`skip_rmsnorm_partials` to add up -- the GEMM has no cross-workgroup merge then.  ``None`` when the shape
    is not served (the caller runs :func:`w4a16_matmul_prepacked`).  ``_unit_loop_engine`` (tests / tuning): the same planes
    from the third-generation body (gemm_w4_v3.hip) instead of the row-group loop (gemm_w4_v4.hip)."""
    from .norm_act import PartialSums
    L.require_cuda(x, packed_weight, packed_scales)
    n, k = packed_weight.shape[0] * 128, packed_weight.shape[1] * 128
    if x.dtype != torch.float16 or x.shape[-1] != k or n % 8:
        return None
    a = _flatten(x, k)
    m = a.shape[0]
    if m < 1 or m > 64:
        return None
    epilogue = 2 | (0x100 if _unit_loop_engine else 0)
    s = L.lib().ll_w4a16_partials_count_ex(m, n, k, int(group_size), epilogue)
    if s < 1:
        return None
    parts = torch.empty((s, m, n), dtype=torch.float32, device=x.device)
    _launch_prepacked(parts.data_ptr(), a, m, n, k, packed_weight, packed_scales, None, group_size,
                      epilogue, "w4a16_matmul_partials")
    return PartialSums(parts, (*x.shape[:-1], n), x.dtype)


def w8_mtiled_supported(m: int, n: int, k: int, fmt: str, group_k: int | None = None) -> bool:
    """Whether a call of ``m`` rows takes the M x N tiled 8-bit engine (gemm_w8_prefill.hip).  ``fmt``: "fp8" / "int8"
    (fp16 activations: w8a16_matmul) or "w8a8" (smoothquant_matmul)."""
    wf = {"fp8": 1, "int8": 2, "w8a8": 3}[fmt]
    return bool(L.lib().ll_w8_mtiled_supported(m, n, k, wf, int(k if group_k is None else min(group_k, k))))


def w8a16_matmul(
    x: torch.Tensor,
    qweight: torch.Tensor,
    scales: torch.Tensor,
    *,
    group_n: int,
    group_k: int,
    bias: torch.Tensor | None = None,
) -> torch.Tensor:
    """``x @ dequant(qweight).T (+ bias)`` with uint8 (fp8-e4m3 bits) or int8 weights and one
    fp32 scale per ``(group_n, group_k)`` weight block."""
    is_fp8 = qweight.dtype == torch.uint8
    if not is_fp8 and qweight.dtype != torch.int8:
        raise ValueError(f"qweight must be uint8 (fp8) or int8, got {qweight.dtype}")
    if x.dtype != torch.float16:
        raise ValueError(f"w8a16 activations must be fp16, got {x.dtype}")
    if qweight.stride(-1) != 1:
        raise ValueError("qweight last dimension must be contiguous")
    n, k = qweight.shape
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    if group_k % 128 != 0 and group_k < k:
        raise ValueError(f"group_k ({group_k}) must be a multiple of 128 unless it covers K")
    L.require_cuda(x, qweight, scales, bias)
    leading = x.shape[:-1]
    a = _flatten(x, k)
    m = a.shape[0]
    if scales.dtype != torch.float32:
        scales = scales.float()
    if scales.dim() == 1:
        scales = scales.unsqueeze(-1)
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    ws, cnt = L.gemm_workspace(x.device, m, n, k)
    L.check(
        L.lib().ll_w8a16_matmul(
            out.data_ptr(), a.data_ptr(), qweight.data_ptr(), scales.data_ptr(), L.ptr(bias), m, n, k,
            int(group_n), int(min(group_k, k)), L.LL_W_FP8E4M3 if is_fp8 else L.LL_W_INT8, a.stride(0),
            qweight.stride(0), scales.stride(0), scales.stride(1), ws.data_ptr(), cnt.data_ptr(),
            L.stream_ptr(),
        ),
        "w8a16_matmul",
    )
    return out.reshape(*leading, n)


def quantize_activations_int8(a: torch.Tensor):
    """Per-token dynamic int8 quantiser of smoothquant_matmul (w8a8.py:34-68):
    ``scale = absmax/127`` (1.0 if 0), ``q = trunc(x/scale)``.  Returns ``(q int8, scale fp32)``."""
    L.require_cuda(a)
    m, k = a.shape
    q = torch.empty((m, k), dtype=torch.int8, device=a.device)
    s = torch.empty((m,), dtype=torch.float32, device=a.device)
    L.check(
        L.lib().ll_quantize_activations_int8(q.data_ptr(), s.data_ptr(), a.data_ptr(), m, k, a.stride(0),
                                             L.stream_ptr()),
        "quantize_activations_int8",
    )
    return q, s


def smoothquant_matmul(
    x: torch.Tensor,
    qweight: torch.Tensor,
    weight_scales: torch.Tensor,
    *,
    bias: torch.Tensor | None = None,
    _return_int32: bool = False,
) -> torch.Tensor:
    """Dynamic per-token W8A8: int8 x int8 -> int32 (exact) -> ``* a_scale[m] * w_scale[n]``.  ``x``: fp16, or rows already
    through the quantiser (``kernels.norm_act.Int8Rows``)."""
    from .norm_act import Int8Rows
    pre = x if isinstance(x, Int8Rows) else None
    if pre is None and x.dtype != torch.float16:
        raise ValueError(f"smoothquant activations must be fp16, got {x.dtype}")
    if qweight.dtype != torch.int8:
        raise ValueError(f"qweight must be int8, got {qweight.dtype}")
    n, k = qweight.shape
    if x.shape[-1] != k:
        raise ValueError(f"x has {x.shape[-1]} cols but weight expects {k}")
    L.require_cuda(qweight, weight_scales, bias)
    leading = x.shape[:-1]
    if pre is not None:
        qa, a_scale = pre.q, pre.scale
        m = qa.shape[0]
    else:
        L.require_cuda(x)
        a = x.reshape(-1, k)
        if a.stride(-1) != 1:
            a = a.contiguous()
        m = a.shape[0]
        qa, a_scale = quantize_activations_int8(a)
    if weight_scales.dim() > 1:
        weight_scales = weight_scales.squeeze(-1)
    weight_scales = weight_scales.float().contiguous()
    if qweight.stride(1) != 1:
        qweight = qweight.contiguous()
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    out = torch.empty((m, n), dtype=torch.float16, device=qa.device)
    acc = torch.empty((m, n), dtype=torch.int32, device=qa.device) if _return_int32 else None
    ws, cnt = L.gemm_workspace(qa.device, m, n, k)
    L.check(
        L.lib().ll_w8a8_matmul(
            out.data_ptr(), qa.data_ptr(), a_scale.data_ptr(), qweight.data_ptr(),
            weight_scales.data_ptr(), L.ptr(bias), m, n, k, qweight.stride(0), L.ptr(acc),
            ws.data_ptr(), cnt.data_ptr(), L.stream_ptr(),
        ),
        "smoothquant_matmul",
    )
    out = out.reshape(*leading, n)
    if _return_int32:
        return out, acc, qa, a_scale
    return out


def _w8a8_planes(x, qweight, max_splits: int):
    """int32 split-K planes of ``x @ qweight.T`` at decode shapes (``ll_dense_partials`` wfmt 3) + the activation scales;
    ``None`` when the engine does not take the call."""
    from .norm_act import Int8Rows
    n, k = qweight.shape
    if qweight.dtype != torch.int8 or qweight.stride(1) != 1 or x.shape[-1] != k or os.environ.get("LL_W8A8_NO_PARTIALS", "0") == "1":
        return None
    if isinstance(x, Int8Rows):
        qa, a_scale = x.q, x.scale
    else:
        if not x.is_cuda or x.dtype != torch.float16:
            return None
        a = x.reshape(-1, k)
        if a.stride(-1) != 1:
            a = a.contiguous()
        if a.shape[0] < 1 or a.shape[0] > 64:
            return None
        qa, a_scale = quantize_activations_int8(a)
    m = qa.shape[0]
    if m < 1 or m > 64:
        return None
    s = L.lib().ll_dense_partials_count(m, n, k, 3, int(max_splits))
    if s < 1:
        return None
    parts = torch.empty((s, m, n), dtype=torch.int32, device=qa.device)
    rc = L.lib().ll_dense_partials(parts.data_ptr(), qa.data_ptr(), qweight.data_ptr(), 0, m, n, k, 1, k, 3, qa.stride(0),
                                   qweight.stride(0), 0, 0, int(max_splits), L.stream_ptr())
    if rc == 0:
        return None
    if rc != s:
        L.check(rc if rc < 0 else -1, "smoothquant_matmul_partials")
    return parts, a_scale


def _w8a8_scales(weight_scales):
    if weight_scales.dim() > 1:
        weight_scales = weight_scales.squeeze(-1)
    return weight_scales.float().contiguous()


def smoothquant_matmul_partials(x, qweight, weight_scales, *, bias=None, max_splits: int = 12):
    """Decode-step extension: :func:`smoothquant_matmul` left as exact int32 split-K planes + scales
    (``ScaledInt32Partials``) for ``skip_rmsnorm_q8`` -- no finish launch.  ``x``: fp16 or ``Int8Rows``.  ``None`` when
    not served (more than 64 rows, shapes off the engine's grid)."""
    from .norm_act import ScaledInt32Partials
    got = _w8a8_planes(x, qweight, min(max_splits, int(os.environ.get("LL_W8A8_SPLITS", max_splits))))
    if got is None:
        return None
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    return ScaledInt32Partials(got[0], (*x.shape[:-1], qweight.shape[0]), got[1], _w8a8_scales(weight_scales), bias)


def smoothquant_gate_up_swiglu(x, qweight, weight_scales, *, max_splits: int = 16):
    """``silu(gate) * up`` of a fused gate|up smoothquant projection whose rows are interleaved ``(gate_j, up_j)``: the int8
    GEMM leaves int32 planes and ONE element-wise launch applies the scale epilogue and the activation (instead of finish +
    swiglu).  Same values as ``swiglu_forward(*smoothquant_matmul(x, ...).unflatten(-1, (-1, 2)).unbind(-1))``.  ``None``
    when not served."""
    n = qweight.shape[0]
    if n % 4:
        return None
    got = _w8a8_planes(x, qweight, max_splits)
    if got is None:
        return None
    parts, a_scale = got
    s, m, _ = parts.shape
    out = torch.empty((m, n // 2), dtype=torch.float16, device=parts.device)
    L.check(L.lib().ll_w8a8_finish_swiglu(out.data_ptr(), parts.data_ptr(), s, a_scale.data_ptr(),
                                          _w8a8_scales(weight_scales).data_ptr(), m, n, L.stream_ptr()),
            "smoothquant_gate_up_swiglu")
    return out.view(*x.shape[:-1], n // 2)


def smoothquant_rows_matmul(x, qweight, weight_scales, *, bias=None, gate_up_swiglu: bool = False):
    """Extension (round 5): :func:`smoothquant_matmul` -- or, with ``gate_up_swiglu``, ``silu(gate) * up`` of a fused gate|up
    projection whose rows are interleaved ``(gate_j, up_j)`` -- for decode shapes with a WIDE output on the row-group loop
    (csrc/gemm_w16_rows.hip, int8 form): ONE launch, exact int32 sums, the reference's scale epilogue, no planes and no finish
    launch.  ``x``: fp16 rows or ``Int8Rows``.  Same values as ``smoothquant_matmul`` (+ ``swiglu_forward``) bit for bit.
    ``None`` when not served (more than 64 rows, n < 8192, k % 256, ``LL_W8A8_ROWS_OFF``)."""
    from .norm_act import Int8Rows
    n, k = qweight.shape
    if (qweight.dtype != torch.int8 or qweight.stride(1) != 1 or x.shape[-1] != k or os.environ.get("LL_W8A8_ROWS_OFF")
            or n < 8192 or (gate_up_swiglu and bias is not None)):
        return None
    if isinstance(x, Int8Rows):
        qa, a_scale = x.q, x.scale
    else:
        if not x.is_cuda or x.dtype != torch.float16:
            return None
        a = x.reshape(-1, k)
        if a.shape[0] < 1 or a.shape[0] > 64:
            return None
        if a.stride(-1) != 1:
            a = a.contiguous()
        qa, a_scale = quantize_activations_int8(a)
    m = qa.shape[0]
    if (not L.lib().ll_w8a8_rows_supported(m, n, k, 1 if gate_up_swiglu else 0) or qa.stride(0) % 16 or qweight.stride(0) % 16
            or qa.data_ptr() % 16 or qweight.data_ptr() % 16):
        return None
    ws = _w8a8_scales(weight_scales)
    if bias is not None and bias.dtype != torch.float16:
        bias = bias.half()
    n_out = n // 2 if gate_up_swiglu else n
    out = torch.empty((m, n_out), dtype=torch.float16, device=qa.device)
    L.check(L.lib().ll_w8a8_rows_matmul(out.data_ptr(), qa.data_ptr(), a_scale.data_ptr(), qweight.data_ptr(), ws.data_ptr(), L.ptr(bias),
                                        m, n, k, qa.stride(0), qweight.stride(0), 1 if gate_up_swiglu else 0, L.stream_ptr()),
            "smoothquant_rows_matmul")
    return out.view(*x.shape[:-1], n_out)


def dense_matmul_partials(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor | None = None, *, group_n: int = 1,
                          group_k: int = 0, max_splits: int = 12):
    """Decode-step extension for the 8-bit and the unquantised 16-bit formats (the int4 route's
    :func:`w4a16_matmul_partials`): the projection ``x @ dequant(weight).T`` left as ``S <= max_splits`` fp32 split-K partial
    planes (:class:`PartialSums`) for its consumer to add up -- no finish launch.  ``weight``: uint8 (fp8 e4m3 bits) / int8
    with ``scales`` (one fp32 per ``group_n x group_k`` block) and fp16 activations, or fp16 / bf16 (``scales`` None,
    activations of the same type).  ``None`` when the call is not served (more than 64 rows, shapes off the engine's grid)."""
    from .norm_act import PartialSums
    if not x.is_cuda or weight.dim() != 2 or weight.stride(1) != 1 or os.environ.get("LL_DENSE_NO_PARTIALS"):
        return None
    n, k = weight.shape
    if weight.dtype in (torch.float16, torch.bfloat16):
        if x.dtype != weight.dtype or scales is not None:
            return None
        wfmt = 4 if weight.dtype == torch.float16 else 5
    elif weight.dtype in (torch.uint8, torch.int8):
        if x.dtype != torch.float16 or scales is None:
            return None
        wfmt = 1 if weight.dtype == torch.uint8 else 2
        if scales.dtype != torch.float32:
            scales = scales.float()
        if scales.dim() == 1:
            scales = scales.unsqueeze(-1)
    else:
        return None
    if x.shape[-1] != k:
        return None
    a = _flatten(x, k)
    m = a.shape[0]
    s = L.lib().ll_dense_partials_count(m, n, k, wfmt, int(max_splits)) if 1 <= m <= 64 else 0
    if s < 1:
        return None
    parts = torch.empty((s, m, n), dtype=torch.float32, device=x.device)
    gk = int(min(group_k, k)) if group_k else k
    rc = L.lib().ll_dense_partials(parts.data_ptr(), a.data_ptr(), weight.data_ptr(), L.ptr(scales), m, n, k, int(group_n), gk, wfmt,
                                   a.stride(0), weight.stride(0), scales.stride(0) if scales is not None else 0,
                                   scales.stride(1) if scales is not None else 0, int(max_splits), L.stream_ptr())
    if rc == 0:
        return None
    if rc < 0:
        L.check(rc, "dense_matmul_partials")
    assert rc == s
    return PartialSums(parts, (*x.shape[:-1], n), x.dtype)


def dense16_wins(m: int, n: int, k: int) -> bool:
    """Where the hand-written kernel measured FASTER than the library GEMM on MI355X (benchmarks/dense16_shapes.py, round 3,
    same box, hipGraph replays over rotating weights): batches of <= 32 rows with a long contraction (Qwen2.5-1.5B down
    8960 -> 1536: 20.4 vs 21.8 us) or a very wide output written straight from the single split (lm_head 151936 x 1536: 94.9
    vs 104.1 us).  Elsewhere hipBLASLt is ahead (7 vs 9 us on the small projections, 211 vs 230 us on the 7B lm_head at
    batch 64) and stays -- a plain library GEMM, the reference's own choice for these layers."""
    return m <= 32 and (k >= 4096 or n >= 32768)


def dense16_rows_wins(m: int, n: int, k: int) -> bool:
    """Where the 16-bit row-group kernel measured faster than the library GEMM and than the split-K 16-bit engine on MI355X
    (benchmarks/dense16_rows.py, round 5, hipGraph replays over rotating weights, us per launch rows / library / split-K):
    fused gate|up 17920 x 1536 at batch 32 incl. the swiglu 13.4 / 25.2 / 30.9; lm_head 151936 x 1536 at batch 32 80.6 / 106.4 /
    96.9 (5.8 TB/s); lm_head 152064 x 3584 at batch 64 204.8 / 210.9 / 226.8; gate|up 37888 x 3584 fp16 at batch 64 55.7 / 76.0 /
    85.3.  It needs about one 32-row group per CU to fill the chip: narrow outputs (down 1536 x 8960: 35.8 / 21.8 / 20.7) keep the
    split-K engine or the library."""
    return 1 <= m <= 64 and n >= 8192 and n % 32 == 0 and k % 128 == 0


def dense16_rows_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, *, gate_up_swiglu: bool = False):
    """Extension (round 5): ``F.linear(x, weight, bias)`` -- or, with ``gate_up_swiglu``, ``silu(gate) * up`` of a fused gate|up
    weight whose rows are interleaved ``(gate_j, up_j)`` -- for decode shapes (<= 64 rows) of an UNQUANTISED fp16 / bf16 weight
    on the row-group weight-streaming kernel (csrc/gemm_w16_rows.hip): finished outputs in one launch, no planes.  ``None`` when
    the call is not served (shape / dtype / alignment, ``LL_DENSE16_ROWS_OFF``): the caller keeps its route."""
    if (not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16) or weight.dtype != x.dtype or weight.dim() != 2
            or os.environ.get("LL_DENSE16_ROWS_OFF")):
        return None
    n, k = weight.shape
    if x.shape[-1] != k or (bias is not None and (bias.dtype != x.dtype or bias.numel() != n or gate_up_swiglu)):
        return None
    a = _flatten(x, k)
    m = a.shape[0]
    if (not L.lib().ll_dense16_rows_supported(m, n, k, 1 if gate_up_swiglu else 0) or weight.stride(1) != 1 or weight.stride(0) % 8
            or weight.data_ptr() % 16 or (m - 1) * a.stride(0) * 2 + k * 2 >= 2 ** 31):
        return None
    n_out = n // 2 if gate_up_swiglu else n
    out = torch.empty((m, n_out), dtype=x.dtype, device=x.device)
    L.check(L.lib().ll_dense16_rows_matmul(out.data_ptr(), a.data_ptr(), weight.data_ptr(), L.ptr(bias), m, n, k, a.stride(0),
                                           weight.stride(0), L.dtype_code(x.dtype), 1 if gate_up_swiglu else 0, L.stream_ptr()),
            "dense16_rows_linear")
    return out.reshape(*x.shape[:-1], n_out)


def dense16_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, policy: str = "always"):
    """Extension: ``F.linear(x, weight, bias)`` for decode shapes (<= 64 rows) of an UNQUANTISED fp16 / bf16 weight
    ``[N, K]`` on the split-K weight-streaming kernel (csrc/gemm_w8_skinny.hip, 16-bit form) -- the reference's
    ``UnquantizedLinearMethod.apply`` / lm_head are torch's library GEMM.  Returns ``None`` when the call is not served
    (other shapes, dtypes, CPU tensors, ``LL_DENSE16_OFF``) or -- ``policy="auto"`` -- where the library measured faster
    (:func:`dense16_wins`): the caller keeps ``F.linear``."""
    if policy == "auto" and x.shape[-1] and not dense16_wins(x.numel() // x.shape[-1], weight.shape[0], weight.shape[1]):
        return None
    if (not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16) or weight.dtype != x.dtype or weight.dim() != 2
            or os.environ.get("LL_DENSE16_OFF")):
        return None
    n, k = weight.shape
    if x.shape[-1] != k or (bias is not None and (bias.dtype != x.dtype or bias.numel() != n)):
        return None
    a = _flatten(x, k)
    m = a.shape[0]
    if m < 1 or m > 64 or k % 64 or n % 4 or weight.stride(1) != 1 or weight.stride(0) % 8 or weight.data_ptr() % 16:
        return None
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    ws, _ = L.gemm_workspace(x.device, m, n, k)
    rc = L.lib().ll_dense16_matmul(out.data_ptr(), a.data_ptr(), weight.data_ptr(), L.ptr(bias), m, n, k, a.stride(0),
                                   weight.stride(0), L.dtype_code(x.dtype), ws.data_ptr(), L.stream_ptr())
    if rc == 0:
        return None
    if rc < 0:
        L.check(rc, "dense16_matmul")
    return out.reshape(*x.shape[:-1], n)
