"""flash_decoding / flash_attention2_no_pad -- mirror of lite_llama/kernels/flashdecoding.py:316-380
and flashattention2_nopad.py:175-231 over the HIP C-ABI."""

from __future__ import annotations

import torch

from .. import _lib as L


# zeroed, self-cleaning counters for the one-launch merge (kernels never allocate; grown outside graph
# capture; outgrown vectors stay alive for graphs that captured their address).  One vector per
# (device, stream) for eager calls, one per device for captures -- see _lib.scratch_keys.
_fd_counters: dict = {}
_fd_keepalive: list = []


def _grow_counters(key, device, entries: int):
    cur = _fd_counters.get(key)
    if cur is not None and cur.numel() >= entries:
        return cur
    if torch.cuda.is_current_stream_capturing():
        return None  # first use inside a capture: take the two-launch form this once
    if cur is not None:
        _fd_keepalive.append(cur)
    cur = torch.zeros(max(entries, 4096), dtype=torch.int32, device=device)
    _fd_counters[key] = cur
    return cur


def _merge_counters(device, entries: int):
    key, cap = L.scratch_keys(device)
    if key != cap:
        _grow_counters(cap, device, entries)
    return _grow_counters(key, device, entries)


@torch.no_grad()
def flash_decoding(
    q,
    k_cache,
    v_cache,
    qk_scale,
    b_req_tokens_table,
    b_req_idx,
    b_seq_len,
    max_actual_seq_len,
):
    """Decode attention for one new token per batch row against the token-attention pool.

    ``k_cache`` / ``v_cache`` may be strided views of ``[max_tokens, 2*Hkv, D]`` (last dim
    contiguous); index tensors may be int32 or int64; the output is a fresh ``[B, Hq, D]``.
    Scratch (``mid_o``, ``mid_o_logexpsum``) is allocated per call like the reference does
    (flashdecoding.py:347-358) -- from torch's caching allocator, so it is graph-safe.
    """
    L.require_cuda(q, k_cache, v_cache, b_req_tokens_table, b_req_idx, b_seq_len)
    assert q.shape[-1] == k_cache.shape[-1] == v_cache.shape[-1]
    assert b_req_tokens_table.dtype == torch.int32
    batchs, num_heads, head_dim = q.shape
    if q.stride(-1) != 1:
        q = q.contiguous()
    assert k_cache.stride(-1) == 1 and v_cache.stride(-1) == 1
    if b_req_tokens_table.stride(1) != 1:
        b_req_tokens_table = b_req_tokens_table.contiguous()
    max_len = int(max_actual_seq_len)
    nparts = L.lib().ll_flash_decoding_num_partitions(max_len)
    mid_o = torch.empty((batchs, num_heads, max(nparts, 1), head_dim), dtype=torch.float32, device=q.device)
    out = torch.empty_like(q, memory_format=torch.contiguous_format)
    n_kv = k_cache.shape[1]
    counters = _merge_counters(q.device, batchs * n_kv * ((num_heads // n_kv + 15) // 16))
    # one-launch form: a cache line (32 floats) per log-sum-exp record
    mid_lse = torch.empty((batchs, num_heads, max(nparts, 1), 1 if counters is None else 32), dtype=torch.float32,
                          device=q.device)
    L.check(
        L.lib().ll_flash_decoding(
            out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
            b_req_tokens_table.data_ptr(), b_req_idx.data_ptr(), b_seq_len.data_ptr(),
            mid_o.data_ptr(), mid_lse.data_ptr(), batchs, num_heads, k_cache.shape[1], head_dim,
            max_len, float(qk_scale), q.stride(0), q.stride(1), k_cache.stride(0), k_cache.stride(1),
            v_cache.stride(0), v_cache.stride(1), out.stride(0), out.stride(1),
            b_req_tokens_table.stride(0), L.dtype_code(q.dtype), L.index_width(b_req_idx),
            L.index_width(b_seq_len), L.ptr(counters), L.stream_ptr(),
        ),
        "flash_decoding",
    )
    return out


@torch.no_grad()
def flash_decoding_fp8kv(q, k_cache, v_cache, qk_scale, b_req_tokens_table, b_req_idx, b_seq_len, max_actual_seq_len,
                         k_scale: float = 1.0, v_scale: float = 1.0):
    """``flash_decoding`` over an fp8 (OCP e4m3) pool written by ``update_kv_buffer_fp8`` (extension): ``k_cache`` /
    ``v_cache`` are uint8 / float8_e4m3fn views ``[rows, Hkv, D]``; fp16 queries, ``D`` 64 or 128."""
    L.require_cuda(q, k_cache, v_cache, b_req_tokens_table, b_req_idx, b_seq_len)
    assert q.dtype == torch.float16 and k_cache.element_size() == 1 and v_cache.element_size() == 1
    assert q.shape[-1] == k_cache.shape[-1] == v_cache.shape[-1] and b_req_tokens_table.dtype == torch.int32
    batchs, num_heads, head_dim = q.shape
    if q.stride(-1) != 1:
        q = q.contiguous()
    assert k_cache.stride(-1) == 1 and v_cache.stride(-1) == 1
    if b_req_tokens_table.stride(1) != 1:
        b_req_tokens_table = b_req_tokens_table.contiguous()
    max_len = int(max_actual_seq_len)
    nparts = L.lib().ll_flash_decoding_num_partitions(max_len)
    mid_o = torch.empty((batchs, num_heads, max(nparts, 1), head_dim), dtype=torch.float32, device=q.device)
    out = torch.empty_like(q, memory_format=torch.contiguous_format)
    n_kv = k_cache.shape[1]
    counters = _merge_counters(q.device, batchs * n_kv * ((num_heads // n_kv + 15) // 16))
    if counters is None:
        raise L.KernelError("flash_decoding_fp8kv needs the merge counters (not available during this capture)")
    mid_lse = torch.empty((batchs, num_heads, max(nparts, 1), 32), dtype=torch.float32, device=q.device)
    L.check(
        L.lib().ll_flash_decoding_fp8kv(
            out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), b_req_tokens_table.data_ptr(),
            b_req_idx.data_ptr(), b_seq_len.data_ptr(), mid_o.data_ptr(), mid_lse.data_ptr(), batchs, num_heads, n_kv,
            head_dim, max_len, float(qk_scale), float(k_scale), float(v_scale), q.stride(0), q.stride(1),
            k_cache.stride(0), k_cache.stride(1), v_cache.stride(0), v_cache.stride(1), out.stride(0), out.stride(1),
            b_req_tokens_table.stride(0), L.index_width(b_req_idx), L.index_width(b_seq_len), L.ptr(counters),
            L.stream_ptr(),
        ),
        "flash_decoding_fp8kv",
    )
    return out


def decode_attention_supported(q, kv, cos, n_kv_heads: int) -> bool:
    """Shapes the one-launch decode attention serves: head_dim >= 64, at most 16 query heads per KV
    head, dense [heads, head_dim] rows, 16-bit tables of the activation dtype."""
    n, hq, hd = q.shape
    return (hd >= 64 and hd % 32 == 0 and hq % n_kv_heads == 0 and hq // n_kv_heads <= 16 and q.stride(2) == 1
            and q.stride(1) == hd and kv.stride(2) == 1 and kv.stride(1) == hd and cos.dtype == q.dtype
            and q.dtype in (torch.float16, torch.bfloat16) and q.stride(0) % 8 == 0 and kv.stride(0) % 8 == 0)


@torch.no_grad()
def _head_norm_ok(qk_norm, head_dim: int, dtype) -> bool:
    if qk_norm is None:
        return True
    qw, kw, _ = qk_norm
    return (head_dim == 128 and all(w.dtype == dtype and w.is_contiguous() and w.numel() == head_dim and w.is_cuda
                                    for w in (qw, kw)))


@torch.no_grad()
def decode_attention(q, kv, cos_table, sin_table, positions, select_index, kv_buffer, qk_scale, b_req_tokens_table,
                     b_req_idx, b_seq_len, max_actual_seq_len, qk_norm=None):
    """Extension: ``rope_and_cache(q, kv, tables, positions)`` + ``flash_decoding(...)`` in ONE launch
    (same values).  ``q [B, Hq, D]`` un-rotated (left untouched), ``kv [B, 2*Hkv, D]`` this step's K
    heads then V heads (left untouched), ``kv_buffer [max_tokens, 2*Hkv, D]`` the layer's pool: row
    ``select_index[b]`` receives the rotated K and the V.  ``qk_norm = (q_weight, k_weight, eps)`` (Qwen3, head_dim 128):
    the per-head RMSNorm of q and of the new K heads runs inside the launch, in front of the rotation, with the values
    of ``skip_rmsnorm`` on the ``[.., D]`` views.  Returns ``None`` when the shape is not
    served or the merge counters cannot be set up (caller runs the two-call form)."""
    L.require_cuda(q, kv, cos_table, sin_table, positions, select_index, kv_buffer, b_req_tokens_table, b_req_idx,
                   b_seq_len)
    batchs, num_heads, head_dim = q.shape
    n_kv = kv.shape[1] // 2
    if not decode_attention_supported(q, kv, cos_table, n_kv) or not _head_norm_ok(qk_norm, head_dim, q.dtype):
        return None
    counters = _merge_counters(q.device, batchs * n_kv)
    positions = positions.reshape(-1)
    if (counters is None or sin_table.dtype != cos_table.dtype or cos_table.dim() != 2 or cos_table.stride(1) != 1
            or sin_table.stride(1) != 1 or cos_table.stride(0) != sin_table.stride(0) or cos_table.stride(0) % 8 != 0
            or positions.dtype != torch.int64 or not positions.is_contiguous() or positions.shape[0] != batchs
            or kv_buffer.stride(2) != 1 or b_req_tokens_table.dtype != torch.int32 or b_req_tokens_table.stride(1) != 1):
        return None
    k_cache, v_cache = kv_buffer[:, :n_kv], kv_buffer[:, n_kv:]
    max_len = int(max_actual_seq_len)
    nparts = L.lib().ll_flash_decoding_num_partitions(max_len)
    mid_o = torch.empty((batchs, num_heads, max(nparts, 1), head_dim), dtype=torch.float32, device=q.device)
    mid_lse = torch.empty((batchs, num_heads, max(nparts, 1), 32), dtype=torch.float32, device=q.device)
    out = torch.empty((batchs, num_heads, head_dim), dtype=q.dtype, device=q.device)
    L.check(
        L.lib().ll_decode_attention(
            out.data_ptr(), q.data_ptr(), kv.data_ptr(), kv.stride(0), cos_table.data_ptr(), sin_table.data_ptr(),
            cos_table.stride(0), positions.data_ptr(), select_index.data_ptr(), L.index_width(select_index),
            k_cache.data_ptr(), v_cache.data_ptr(), b_req_tokens_table.data_ptr(), b_req_idx.data_ptr(),
            b_seq_len.data_ptr(), mid_o.data_ptr(), mid_lse.data_ptr(), batchs, num_heads, n_kv, head_dim, max_len,
            float(qk_scale), q.stride(0), q.stride(1), k_cache.stride(0), k_cache.stride(1), v_cache.stride(0),
            v_cache.stride(1), out.stride(0), out.stride(1), b_req_tokens_table.stride(0), L.dtype_code(q.dtype),
            L.index_width(b_req_idx), L.index_width(b_seq_len), counters.data_ptr(),
            L.ptr(None if qk_norm is None else qk_norm[0]), L.ptr(None if qk_norm is None else qk_norm[1]),
            0.0 if qk_norm is None else float(qk_norm[2]), L.stream_ptr(),
        ),
        "decode_attention",
    )
    return out


def decode_attention_partials_supported(max_actual_seq_len, num_heads: int, num_kv_heads: int, head_dim: int, batch: int = 1) -> bool:
    """Whether :func:`decode_attention_partials` serves this step (known before the projection is launched): the
    one-workgroup-per-(row, KV head) form -- contexts of 129 .. 1024 tokens at any batch, longer ones (round 6: the waves walk the
    partitions) when ``batch x KV heads`` alone fills half the chip."""
    waves = L.lib().ll_flash_decoding_group_waves(int(max_actual_seq_len), int(batch) * int(num_kv_heads))
    return (waves >= 2 and head_dim >= 64 and head_dim % 32 == 0 and num_heads % num_kv_heads == 0
            and num_heads // num_kv_heads <= 16 and (num_heads // num_kv_heads + 2) * head_dim // 4 <= 128 * waves)


@torch.no_grad()
def decode_attention_partials(parts, bias, num_heads: int, num_kv_heads: int, head_dim: int, cos_table, sin_table,
                              positions, select_index, kv_buffer, qk_scale, b_req_tokens_table, b_req_idx, b_seq_len,
                              max_actual_seq_len, qk_norm=None):
    """:func:`decode_attention` whose ``q | k | v`` come as the fp32 split-K partials of the fused projection
    (``PartialSums`` ``[S, B, (Hq + 2 Hkv) D]``, ``bias`` the projection bias or ``None``): every workgroup adds the
    partials of its (row, KV head) -- plus bias, one rounding to the pool dtype: the value the projection itself would
    have stored -- and continues as ``decode_attention`` (``qk_norm`` as there).  ``parts`` may also be the
    ``ScaledInt32Partials`` of a smoothquant projection (fp16, head_dim 128): the launch applies the scale epilogue.  Returns ``None`` when the shape is not served (contexts of
    129..1024 tokens -- any longer one when batch x KV heads >= 128 workgroups --, head_dim >= 64, <= 16 query heads per KV head):
    finish the sums and call ``decode_attention``."""
    p = parts.parts
    scaled = p.dtype == torch.int32  # smoothquant: exact int32 planes + per-token / per-channel scales (ScaledInt32Partials)
    if scaled and (qk_norm is not None or head_dim != 128 or kv_buffer.dtype != torch.float16
                   or getattr(parts, "a_scale", None) is None):
        return None
    L.require_cuda(p, bias, cos_table, sin_table, positions, select_index, kv_buffer, b_req_tokens_table, b_req_idx, b_seq_len)
    s_count, batchs, row_w = p.shape
    dt = kv_buffer.dtype
    positions = positions.reshape(-1)
    if ((p.dtype != torch.float32 and not scaled) or row_w != (num_heads + 2 * num_kv_heads) * head_dim or head_dim < 64 or head_dim % 32 or num_heads % num_kv_heads
            or num_heads // num_kv_heads > 16 or s_count > 8 or not p.is_contiguous() or parts.dtype != dt
            or dt not in (torch.float16, torch.bfloat16) or cos_table.dtype != dt or sin_table.dtype != dt
            or cos_table.dim() != 2 or cos_table.stride(1) != 1 or sin_table.stride(1) != 1
            or cos_table.stride(0) != sin_table.stride(0) or cos_table.stride(0) % 8 != 0
            or positions.dtype != torch.int64 or not positions.is_contiguous() or positions.shape[0] != batchs
            or kv_buffer.stride(2) != 1 or b_req_tokens_table.dtype != torch.int32 or b_req_tokens_table.stride(1) != 1
            or (bias is not None and (bias.dtype != dt or not bias.is_contiguous() or bias.numel() != row_w))
            or not _head_norm_ok(qk_norm, head_dim, dt)):
        return None
    max_len = int(max_actual_seq_len)
    waves = L.lib().ll_flash_decoding_group_waves(max_len, batchs * num_kv_heads)  # the one-workgroup form (0: not applicable)
    if waves < 2 or (num_heads // num_kv_heads + 2) * head_dim // 4 > 128 * waves:
        return None
    k_cache, v_cache = kv_buffer[:, :num_kv_heads], kv_buffer[:, num_kv_heads:]
    out = torch.empty((batchs, num_heads, head_dim), dtype=dt, device=p.device)
    L.check(
        L.lib().ll_decode_attention_partials(
            out.data_ptr(), p.data_ptr(), s_count, L.ptr(bias), cos_table.data_ptr(), sin_table.data_ptr(),
            cos_table.stride(0), positions.data_ptr(), select_index.data_ptr(), L.index_width(select_index),
            k_cache.data_ptr(), v_cache.data_ptr(), b_req_tokens_table.data_ptr(), b_req_idx.data_ptr(),
            b_seq_len.data_ptr(), batchs, num_heads, num_kv_heads, head_dim, max_len, float(qk_scale),
            k_cache.stride(0), k_cache.stride(1), v_cache.stride(0), v_cache.stride(1), out.stride(0), out.stride(1),
            b_req_tokens_table.stride(0), L.dtype_code(dt), L.index_width(b_req_idx), L.index_width(b_seq_len),
            L.ptr(None if qk_norm is None else qk_norm[0]), L.ptr(None if qk_norm is None else qk_norm[1]),
            0.0 if qk_norm is None else float(qk_norm[2]), parts.a_scale.data_ptr() if scaled else 0,
            parts.w_scale.data_ptr() if scaled else 0, L.stream_ptr(),
        ),
        "decode_attention_partials",
    )
    return out


@torch.no_grad()
def flash_attention2_no_pad(q, k, v, sm_scale, b_start_loc, b_seq_len, max_seq_len):
    """Varlen causal prefill attention (``sm_scale`` must already include log2(e); the
    kernel evaluates exp2).  fp32 inputs are cast to fp16 like the reference's
    ``custom_fwd(cast_inputs=torch.float16)``."""
    L.require_cuda(q, k, v, b_start_loc, b_seq_len)
    if q.dtype == torch.float32:
        q, k, v = q.half(), k.half(), v.half()
    if q.stride(-1) != 1:
        q = q.contiguous()
    if k.stride(-1) != 1:
        k = k.contiguous()
    if v.stride(-1) != 1:
        v = v.contiguous()
    output = torch.empty_like(q, memory_format=torch.contiguous_format)
    batchs = b_seq_len.shape[0]
    n_heads, head_dim = q.shape[1], q.shape[2]
    L.check(
        L.lib().ll_flash_attention_nopad(
            output.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), b_start_loc.data_ptr(),
            b_seq_len.data_ptr(), batchs, n_heads, k.shape[1], head_dim, int(max_seq_len),
            float(sm_scale), q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0),
            v.stride(1), output.stride(0), output.stride(1), L.dtype_code(q.dtype),
            L.index_width(b_start_loc), L.index_width(b_seq_len), L.stream_ptr(),
        ),
        "flash_attention2_no_pad",
    )
    return output
