"""update_kv_buffer / update_kv_index -- mirror of lite_llama/kernels/update_kv_buffer.py:54-89
and update_kv_index.py:50-88 over the HIP C-ABI (bit-exact integer / byte moves)."""

from __future__ import annotations

import torch

from .. import _lib as L


@torch.no_grad()
def update_kv_buffer(KV_Values, Select_Index, KV_Buffer):
    """``KV_Buffer[Select_Index[i], :, :] = KV_Values[i, :, :]``; other rows untouched."""
    L.require_cuda(KV_Values, Select_Index, KV_Buffer)
    assert KV_Values.shape[1] == KV_Buffer.shape[1] and KV_Values.shape[2] == KV_Buffer.shape[2]
    assert KV_Values.dtype == KV_Buffer.dtype and KV_Values.element_size() == 2
    if KV_Values.stride(2) != 1:
        KV_Values = KV_Values.contiguous()
    assert KV_Buffer.stride(2) == 1
    L.check(
        L.lib().ll_update_kv_buffer(
            KV_Values.data_ptr(), Select_Index.data_ptr(), KV_Buffer.data_ptr(),
            Select_Index.shape[0], KV_Values.shape[1], KV_Values.shape[2], KV_Values.stride(0),
            KV_Values.stride(1), KV_Buffer.stride(0), KV_Buffer.stride(1),
            L.index_width(Select_Index), L.stream_ptr(),
        ),
        "update_kv_buffer",
    )
    return


FP8_KV_DTYPES = (torch.uint8, torch.float8_e4m3fn)


@torch.no_grad()
def update_kv_buffer_fp8(KV_Values, Select_Index, KV_Buffer, num_k_heads: int, k_scale: float = 1.0, v_scale: float = 1.0):
    """``update_kv_buffer`` into an fp8 (OCP e4m3) pool (extension): ``KV_Buffer[Select_Index[i]] = e4m3(KV_Values[i] /
    scale)`` with ``k_scale`` for the first ``num_k_heads`` heads of a row and ``v_scale`` for the rest; values are
    clamped to +-448 and rounded to nearest even.  ``KV_Buffer``: uint8 or float8_e4m3fn ``[rows, heads, hd]``."""
    L.require_cuda(KV_Values, Select_Index, KV_Buffer)
    assert KV_Values.shape[1] == KV_Buffer.shape[1] and KV_Values.shape[2] == KV_Buffer.shape[2]
    assert KV_Buffer.dtype in FP8_KV_DTYPES and KV_Values.dtype in (torch.float16, torch.bfloat16)
    if KV_Values.stride(2) != 1:
        KV_Values = KV_Values.contiguous()
    assert KV_Buffer.stride(2) == 1
    L.check(
        L.lib().ll_update_kv_buffer_fp8(
            KV_Values.data_ptr(), Select_Index.data_ptr(), KV_Buffer.data_ptr(), Select_Index.shape[0], KV_Values.shape[1],
            num_k_heads, KV_Values.shape[2], KV_Values.stride(0), KV_Values.stride(1), KV_Buffer.stride(0),
            KV_Buffer.stride(1), float(k_scale), float(v_scale), L.dtype_code(KV_Values.dtype),
            L.index_width(Select_Index), L.stream_ptr(),
        ),
        "update_kv_buffer_fp8",
    )


@torch.no_grad()
def update_kv_index(req_to_token_indexs, b_req_idx, b_seq_len, select_index):
    """``table[b_req_idx[i], b_seq_len[i] - 1] = select_index[i]``."""
    L.require_cuda(req_to_token_indexs, b_req_idx, b_seq_len, select_index)
    assert (
        b_seq_len.shape[0] == select_index.shape[0] and b_req_idx.shape[0] == b_seq_len.shape[0]
    ), "b_req_idx, b_seq_len and select_index must have the same length"
    assert req_to_token_indexs.dtype == torch.int32
    L.check(
        L.lib().ll_update_kv_index(
            req_to_token_indexs.data_ptr(), b_req_idx.data_ptr(), b_seq_len.data_ptr(),
            select_index.data_ptr(), b_seq_len.shape[0], req_to_token_indexs.stride(0),
            req_to_token_indexs.stride(1), L.index_width(b_req_idx), L.index_width(b_seq_len),
            L.index_width(select_index), L.stream_ptr(),
        ),
        "update_kv_index",
    )
    return
