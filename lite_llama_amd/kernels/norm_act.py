"""skip_rmsnorm / swiglu_forward / rope_emb_forward -- Python mirror of the reference
wrappers (lite_llama/kernels/skip_rmsnorm.py:192-234, swiglu.py:45-65, rope_emb.py:86-134)
over the HIP C-ABI.  Same names, argument meaning, aliasing and error behaviour."""

from __future__ import annotations

import os

import torch

from .. import _lib as L

MAX_FUSED_SIZE = 65536  # reference kernels/utils.py:20


def _check_row(n: int) -> None:
    # reference calculate_settings (kernels/utils.py:48-54)
    block = 1 << max(0, (n - 1).bit_length())
    if block > MAX_FUSED_SIZE:
        raise RuntimeError(
            f"Cannot launch Triton kernel since n = {n} exceeds "
            f"the recommended Triton blocksize = {MAX_FUSED_SIZE}."
        )


@torch.no_grad()
def skip_rmsnorm(X, residual, weight, eps=1e-5):
    """``(Y, residual)``: residual is updated IN PLACE with ``x + residual`` and returned
    (same storage); ``residual=None`` -> plain RMSNorm, returns ``(Y, X)``."""
    L.require_cuda(X, residual, weight)
    orig_shape = X.shape
    X = X.contiguous().view(-1, orig_shape[-1])
    M, N = X.shape
    _check_row(N)
    Y = torch.empty_like(X)
    if residual is not None:
        residual = residual.contiguous().view(-1, N)
    if weight.dtype != X.dtype:
        weight = weight.to(X.dtype)
    L.check(
        L.lib().ll_skip_rmsnorm(
            Y.data_ptr(), X.data_ptr(), L.ptr(residual), weight.contiguous().data_ptr(), M, N,
            float(eps), L.dtype_code(X.dtype), L.stream_ptr(),
        ),
        "skip_rmsnorm",
    )
    if residual is not None:
        return Y.view(orig_shape), residual.view(orig_shape)
    return Y.view(orig_shape), X.view(orig_shape)


class PartialSums:
    """A projection output left as ``S`` fp32 split-K partial sums ``[S, rows, n]`` (decode-step extension, TP = 1
    only): the W4A16 decode GEMM then has no cross-workgroup merge; the sums are added by the consumer of the
    projection, :func:`skip_rmsnorm_partials`.  ``materialise()`` gives the tensor the projection would have
    returned."""

    __slots__ = ("parts", "shape", "dtype", "tp_reduce")

    def __init__(self, parts: torch.Tensor, shape, dtype, tp_reduce: bool = False):
        """``tp_reduce``: the sums are this RANK's share of a row-parallel projection -- the consumer adds the ranks'
        shares too (fused one-shot all-reduce, csrc/tp_allreduce.hip::allreduce_norm_partials_kernel)."""
        self.parts, self.shape, self.dtype, self.tp_reduce = parts, tuple(shape), dtype, tp_reduce

    def materialise(self) -> torch.Tensor:
        out = self.parts.sum(0).to(self.dtype).view(self.shape)
        if self.tp_reduce:
            from ..distributed.parallel_state import all_reduce_tp
            out = all_reduce_tp(out.contiguous())
        return out


class SlotSums(PartialSums):
    """:class:`PartialSums` of a fused-MoE block: ``parts`` are the down projection's per-slot rows ``[rows, k, n]`` in the
    activation dtype (router weights folded in); their fp32 sum over ``k``, rounded once, is the block's output (``moe_sum``).
    Consumed by :func:`skip_rmsnorm_partials` (one launch instead of moe_sum + norm)."""

    __slots__ = ()

    def __init__(self, parts: torch.Tensor, shape):
        super().__init__(parts, shape, parts.dtype)

    def materialise(self) -> torch.Tensor:
        rows, k, n = self.parts.shape
        out = torch.empty((rows, n), dtype=self.dtype, device=self.parts.device)
        L.check(L.lib().ll_moe_sum(out.data_ptr(), self.parts.data_ptr(), rows, k, n, L.dtype_code(self.dtype), L.stream_ptr()),
                "moe_sum")
        return out.view(self.shape)


class Int8Rows:
    """Activations already through smoothquant's per-token quantiser (``quantize_activations_int8``): ``q`` int8
    ``[rows, k]``, ``scale`` fp32 ``[rows]``; ``shape`` is the logical (fp16) tensor's.  Produced by
    :func:`skip_rmsnorm_q8` (the quantiser fused into the norm launch) and accepted by ``smoothquant_matmul`` and the
    W8A8 linear method in place of the fp16 tensor (decode-step extension; reference w8a8.py:34-68 runs the quantiser as a
    launch of its own in front of every projection)."""

    __slots__ = ("q", "scale", "shape")
    dtype = torch.float16
    is_cuda = True

    def __init__(self, q: torch.Tensor, scale: torch.Tensor, shape):
        self.q, self.scale, self.shape = q, scale, tuple(shape)

    @property
    def device(self):
        return self.q.device

    def view(self, *shape):
        """Only re-groupings of the leading dimensions (the rows stay rows)."""
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        if shape[-1] != self.shape[-1]:
            raise ValueError("Int8Rows.view keeps the last dimension")
        lead = shape[:-1]
        if -1 in lead:
            known = 1
            for d in lead:
                known *= d if d != -1 else 1
            lead = tuple(self.q.shape[0] // known if d == -1 else d for d in lead)
        return Int8Rows(self.q, self.scale, (*lead, shape[-1]))

    def numel(self):
        return self.q.numel()


class ScaledInt32Partials(PartialSums):
    """:class:`PartialSums` of a smoothquant projection: ``parts`` are EXACT int32 split-K sums ``[S, rows, n]`` that still
    need ``* a_scale[m] * w_scale[n] (+ bias)`` -- consumed by :func:`skip_rmsnorm_q8`."""

    __slots__ = ("a_scale", "w_scale", "bias")

    def __init__(self, parts, shape, a_scale, w_scale, bias=None):
        super().__init__(parts, shape, torch.float16)
        self.a_scale, self.w_scale, self.bias = a_scale, w_scale, bias

    def materialise(self) -> torch.Tensor:
        acc = self.parts.sum(0, dtype=torch.int32).float()
        out = (acc * self.a_scale[:, None]) * self.w_scale[None, :]
        if self.bias is not None:
            out = out + self.bias.float()
        return out.to(self.dtype).view(self.shape)


@torch.no_grad()
def skip_rmsnorm_q8(X, residual, weight, eps=1e-5, *, quantize: bool = True, keep_y: bool = False):
    """:func:`skip_rmsnorm` of a smoothquant block in ONE launch (csrc/w8a8_fused.hip): ``X`` is an fp16 tensor or the
    :class:`ScaledInt32Partials` of the previous W8A8 projection (its scale epilogue runs here); with ``quantize`` the
    normalised rows leave as :class:`Int8Rows` for the next W8A8 projection -- the values of ``dense8_finish`` ->
    ``skip_rmsnorm`` -> ``quantize_activations_int8``, bit for bit.  Returns ``(Int8Rows | Y, residual)``; ``keep_y`` also
    stores the fp16 ``Y`` and returns ``((Int8Rows, Y), residual)``."""
    planes = X if isinstance(X, ScaledInt32Partials) else None
    if planes is not None:
        s, m, n = X.parts.shape
        shape = X.shape
        dev = X.parts.device
        if residual is None:
            raise ValueError("skip_rmsnorm_q8 over partials needs a residual")
    else:
        if X.dtype != torch.float16:
            raise ValueError("skip_rmsnorm_q8 is the fp16 smoothquant route")
        shape = X.shape
        X = X.contiguous().view(-1, shape[-1])
        m, n = X.shape
        s, dev = 0, X.device
    _check_row(n)
    if n % 8 or n > 8192:
        raise ValueError(f"skip_rmsnorm_q8: row width {n} outside the fused launch (n % 8 == 0, n <= 8192)")
    if residual is not None:
        residual = residual.contiguous().view(-1, n)
        if residual.shape[0] != m or residual.dtype != torch.float16:
            raise ValueError("residual must be fp16 [rows, n]")
    weight = weight.to(torch.float16).contiguous()
    L.require_cuda(weight, residual)
    want_y = keep_y or not quantize
    Y = torch.empty((m, n), dtype=torch.float16, device=dev) if want_y else None
    q = torch.empty((m, n), dtype=torch.int8, device=dev) if quantize else None
    qs = torch.empty((m,), dtype=torch.float32, device=dev) if quantize else None
    L.check(
        L.lib().ll_skip_rmsnorm_q8(L.ptr(Y), L.ptr(q), L.ptr(qs), 0 if planes is not None else X.data_ptr(),
                                   planes.parts.data_ptr() if planes is not None else 0, s,
                                   planes.a_scale.data_ptr() if planes is not None else 0,
                                   planes.w_scale.data_ptr() if planes is not None else 0,
                                   L.ptr(planes.bias) if planes is not None else 0, L.ptr(residual), weight.data_ptr(), m, n,
                                   float(eps), L.stream_ptr()),
        "skip_rmsnorm_q8",
    )
    res_out = residual.view(shape) if residual is not None else X.view(shape)
    rows = Int8Rows(q, qs, shape) if quantize else None
    if quantize and keep_y:
        return (rows, Y.view(shape)), res_out
    return (rows if quantize else Y.view(shape)), res_out


@torch.no_grad()
def skip_rmsnorm_partials(X: PartialSums, residual, weight, eps=1e-5):
    """:func:`skip_rmsnorm` over a :class:`PartialSums` input: ``x = fp16(sum of the partials)`` -- the value the
    projection itself would have stored -- then the same add-and-normalise.  ``residual`` is required.
    (Round 3's in-launch form -- the consuming projection normalising inside its own launch -- measured 10 us slower per
    pair and was removed in round 4: DESIGN_NOTEBOOK.md 4.2.)"""
    if residual is None:
        raise ValueError("skip_rmsnorm_partials needs a residual (the projection follows a normalised block)")
    L.require_cuda(X.parts, residual, weight)
    if isinstance(X, SlotSums):
        m, s, n = X.parts.shape
    else:
        s, m, n = X.parts.shape
    _check_row(n)
    residual = residual.contiguous().view(-1, n)
    if residual.shape[0] != m or residual.dtype != X.dtype:
        raise ValueError("residual must be [rows, n] in the projection's dtype")
    if weight.dtype != X.dtype:
        weight = weight.to(X.dtype)
    Y = torch.empty((m, n), dtype=X.dtype, device=residual.device)
    if isinstance(X, SlotSums):
        L.check(L.lib().ll_skip_rmsnorm_slots(Y.data_ptr(), X.parts.data_ptr(), s, residual.data_ptr(),
                                              weight.contiguous().data_ptr(), m, n, float(eps), L.dtype_code(X.dtype),
                                              L.stream_ptr()), "skip_rmsnorm_slots")
        return Y.view(X.shape), residual.view(X.shape)
    if X.tp_reduce:
        # tensor parallelism: the partials are this rank's share; one launch adds them, exchanges the fp16 sums with the
        # peers (one-shot all-reduce over peer-mapped buffers, fp32 adds in rank order) and normalises
        from ..distributed import parallel_state as ps
        ps.all_reduce_norm_partials(X.parts, residual, weight.contiguous(), eps, Y)
        return Y.view(X.shape), residual.view(X.shape)
    L.check(
        L.lib().ll_skip_rmsnorm_partials(Y.data_ptr(), X.parts.data_ptr(), s, residual.data_ptr(),
                                         weight.contiguous().data_ptr(), m, n, float(eps), L.dtype_code(X.dtype),
                                         L.stream_ptr()),
        "skip_rmsnorm_partials",
    )
    return Y.view(X.shape), residual.view(X.shape)


def swiglu_forward(a, b):
    """``silu(a.float()) * b`` in ``a``'s dtype and shape."""
    L.require_cuda(a, b)
    ori_shape = a.shape
    n_cols = ori_shape[-1]
    _check_row(n_cols)
    if (a.dim() >= 2 and a.dtype == b.dtype and a.shape == b.shape and a.stride() == b.stride()
            and a.stride(-1) == 1 and b.data_ptr() == a.data_ptr() + n_cols * a.element_size()
            and all(a.stride(i) == a.stride(i + 1) * a.shape[i + 1] for i in range(a.dim() - 2))
            and a.stride(-2) == 2 * n_cols and n_cols % 8 == 0 and a.data_ptr() % 16 == 0):
        # a | b are the two column halves of one row-major [rows, 2n] buffer (the fused gate/up
        # projection): read them in place instead of materialising contiguous copies
        rows = a.numel() // n_cols
        c = torch.empty(ori_shape, dtype=a.dtype, device=a.device)
        L.check(L.lib().ll_silu_and_mul(c.data_ptr(), a.data_ptr(), rows, n_cols, L.dtype_code(a.dtype),
                                        L.stream_ptr()), "swiglu_forward")
        return c
    if (a.dim() >= 2 and a.dtype == b.dtype and a.shape == b.shape and a.stride() == b.stride() and a.stride(-1) == 2
            and b.data_ptr() == a.data_ptr() + a.element_size() and a.stride(-2) == 2 * n_cols
            and all(a.stride(i) == a.stride(i + 1) * a.shape[i + 1] for i in range(a.dim() - 2)) and a.data_ptr() % 16 == 0):
        # a / b are the even / odd columns of one row-major [rows, 2n] buffer: a fused gate|up projection over row-interleaved
        # weights called with more rows than its fused launch serves (prefill) -- read the pairs in place
        rows = a.numel() // n_cols
        c = torch.empty(ori_shape, dtype=a.dtype, device=a.device)
        L.check(L.lib().ll_silu_and_mul_pairs(c.data_ptr(), a.data_ptr(), rows, n_cols, L.dtype_code(a.dtype), L.stream_ptr()),
                "swiglu_forward")
        return c
    a2 = a.reshape(-1, n_cols)
    b2 = b.reshape(-1, n_cols)
    if not a2.is_contiguous():
        a2 = a2.contiguous()
    if not b2.is_contiguous():
        b2 = b2.contiguous()
    c = torch.empty_like(a2)
    L.check(
        L.lib().ll_swiglu(c.data_ptr(), a2.data_ptr(), b2.data_ptr(), a2.shape[0], n_cols,
                          L.dtype_code(a.dtype), L.stream_ptr()),
        "swiglu_forward",
    )
    return c.view(*ori_shape)


def rope_emb_forward(q, k, cos, sin, batch_size, seq_len):
    """In-place half-split RoPE on ``q [N, Hq, D]`` and ``k [N, Hk, D]``; token ``i`` uses
    ``cos[i // seq_len, i % seq_len, : D/2]``.  Returns ``(q, k)`` (the same tensors when
    the inputs are contiguous, exactly like the reference's ``.contiguous()`` calls)."""
    L.require_cuda(q, k, cos, sin)
    N, n_qh, hd = q.shape
    _, n_kh, _ = k.shape
    assert batch_size * seq_len == N

    def _rows_ok(t):  # [N, H, D] with dense heads; the token stride is free (views of a fused qkv row)
        return t.stride(2) == 1 and t.stride(1) == t.shape[2] and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0

    if not _rows_ok(q):
        q = q.contiguous()
    if not _rows_ok(k):
        k = k.contiguous()
    cos = cos.contiguous()
    sin = sin.contiguous()
    if cos.dtype != sin.dtype:
        sin = sin.to(cos.dtype)
    L.check(
        L.lib().ll_rope(
            q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), N, n_qh, n_kh, hd,
            q.stride(0), k.stride(0), seq_len, cos.stride(0), cos.stride(1), sin.stride(0),
            sin.stride(1), L.dtype_code(q.dtype), L.dtype_code(cos.dtype), L.stream_ptr(),
        ),
        "rope_emb_forward",
    )
    return q, k


@torch.no_grad()
def rope_and_cache(q, kv, cos, sin, batch_size, seq_len, select_index, kv_buffer, positions=None):
    """Extension (one launch instead of two): ``rope_emb_forward(q, k)`` + ``update_kv_buffer(cat(k, v))``
    where ``kv [N, 2*Hkv, D]`` holds this step's K heads then V heads (typically a strided view of the
    fused qkv projection).  ``q`` and the K half of ``kv`` are rotated in place; rotated K and V of
    token ``i`` are written to ``kv_buffer[select_index[i]]``.  Bit-identical to the two-call form.
    With ``positions`` (int64 ``[N]``) ``cos``/``sin`` are position-indexed tables ``[max_pos, >= D/2]``
    and token ``i`` reads row ``positions[i]`` (no per-step table producer)."""
    L.require_cuda(q, kv, cos, sin, select_index, kv_buffer)
    N, n_qh, hd = q.shape
    n_kh = kv.shape[1] // 2
    assert batch_size * seq_len == N and kv.shape == (N, 2 * n_kh, hd)
    assert kv_buffer.shape[1] == 2 * n_kh and kv_buffer.shape[2] == hd and kv_buffer.dtype == q.dtype == kv.dtype
    for t in (q, kv):
        if t.stride(2) != 1 or t.stride(1) != hd:
            raise ValueError("rope_and_cache needs dense [heads, head_dim] rows (only the token stride is free)")
    assert kv_buffer.stride(2) == 1
    if cos.dtype != sin.dtype:
        sin = sin.to(cos.dtype)
    if positions is not None:
        L.require_cuda(positions)
        positions = positions.reshape(-1)
        if positions.dtype != torch.int64 or positions.shape[0] != N or not positions.is_contiguous():
            raise ValueError("positions must be a contiguous int64 tensor with one entry per token")
        if cos.dim() != 2 or sin.dim() != 2 or cos.stride(1) != 1 or sin.stride(1) != 1:
            raise ValueError("with positions, cos / sin must be [max_pos, >= head_dim/2] tables")
        strides = (0, cos.stride(0), 0, sin.stride(0))
    else:
        cos = cos.contiguous()
        sin = sin.contiguous()
        strides = (cos.stride(0), cos.stride(1), sin.stride(0), sin.stride(1))
    L.check(
        L.lib().ll_rope_kv_update(
            q.data_ptr(), kv.data_ptr(), cos.data_ptr(), sin.data_ptr(), kv_buffer.data_ptr(),
            select_index.data_ptr(), N, n_qh, n_kh, hd, q.stride(0), kv.stride(0), seq_len, *strides,
            kv_buffer.stride(0), kv_buffer.stride(1), L.dtype_code(q.dtype), L.dtype_code(cos.dtype),
            L.index_width(select_index), L.ptr(positions), L.stream_ptr(),
        ),
        "rope_and_cache",
    )
    return q, kv
