"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the AutoAWQ (0.2.x, GEMM) and AutoGPTQ (0.7.x,
incl. the act-order meaning of ``g_idx``) int4 checkpoint layouts: their packers (to build test checkpoints) and the conversion
into the reference's native W4A16 layout (lite_llama/kernels/quantization/w4a16.py:152-207,
models/quantization/params/int4.py:33-49).

PARITY UNPINNED against third-party code: neither library is in /root/reference or this image (the
reference cannot load such checkpoints, models/weights.py:166-173,266-268).  What IS pinned: the
conversion target -- the native layout -- through ``oracle.dequant_int4`` (itself pinned by the
reference-generated ``quantize_int4_groupwise`` / ``w4a16_*`` fixtures), and hand-computed
known-answer words in tests/test_w4_layouts.py.
"""

from __future__ import annotations

import numpy as np

AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)  # AutoAWQ WQLinear_GEMM.from_linear: order_map


def _pack_cols(vals: np.ndarray, order) -> np.ndarray:
    """[R, C] values 0..15 -> [R, C/8] uint32; nibble i of word j holds column 8j + order[i]."""
    r, c = vals.shape
    v = vals.astype(np.uint32).reshape(r, c // 8, 8)
    out = np.zeros((r, c // 8), dtype=np.uint32)
    for i, src in enumerate(order):
        out |= v[:, :, src] << np.uint32(4 * i)
    return out


def _unpack_cols(words: np.ndarray, order) -> np.ndarray:
    r, cw = words.shape
    out = np.zeros((r, cw, 8), dtype=np.uint32)
    for i, dst in enumerate(order):
        out[:, :, dst] = (words.astype(np.uint32) >> np.uint32(4 * i)) & 0xF
    return out.reshape(r, cw * 8)


def awq_pack(q_kn: np.ndarray, zeros_gn: np.ndarray):
    """q [K, N] and z [K/g, N] (0..15) -> (qweight int32 [K, N/8], qzeros int32 [K/g, N/8])."""
    return _pack_cols(q_kn, AWQ_ORDER).view(np.int32), _pack_cols(zeros_gn, AWQ_ORDER).view(np.int32)


def gptq_pack(q_kn: np.ndarray, zeros_gn: np.ndarray, v1: bool = True):
    """q [K, N], z [K/g, N] -> (qweight int32 [K/8, N] packed along K, qzeros int32 [K/g, N/8]
    sequential, holding ``z - 1`` (low 4 bits) for v1 checkpoints)."""
    k, n = q_kn.shape
    qw = _pack_cols(q_kn.T.copy(), range(8)).T.copy()          # [K/8, N]
    stored = (zeros_gn.astype(np.int64) - (1 if v1 else 0)) & 0xF
    return qw.view(np.int32), _pack_cols(stored, range(8)).view(np.int32)


def native_pack(q_nk: np.ndarray) -> np.ndarray:
    """[N, K] values -> the reference's qweight [N, K/8] int32 (sequential along K, LSB first)."""
    return _pack_cols(q_nk, range(8)).view(np.int32)


def awq_to_native(qweight, qzeros, scales_f16, group_size):
    q_kn = _unpack_cols(qweight.view(np.uint32), AWQ_ORDER)
    z_gn = _unpack_cols(qzeros.view(np.uint32), AWQ_ORDER)
    return native_pack(q_kn.T.copy()), scales_f16.astype(np.float32).T.copy(), z_gn.astype(np.float32).T.copy()


def gptq_to_native(qweight, qzeros, scales_f16, group_size, v1: bool = True):
    q_nk = _unpack_cols(qweight.view(np.uint32).T.copy(), range(8))   # [N, K]
    z_gn = _unpack_cols(qzeros.view(np.uint32), range(8)) + (1 if v1 else 0)
    return native_pack(q_nk), scales_f16.astype(np.float32).T.copy(), z_gn.astype(np.float32).T.copy()


def dequant_kn(q_kn, zeros_gn, scales_gn_f16, group_size):
    """The checkpoint's own meaning: W[k, n] = (q - z[k // g]) * s[k // g] in fp32 -> [K, N]."""
    z = np.repeat(zeros_gn.astype(np.float32), group_size, axis=0)
    s = np.repeat(scales_gn_f16.astype(np.float32), group_size, axis=0)
    return (q_kn.astype(np.float32) - z) * s


def dequant_kn_act_order(q_kn, zeros_gn, scales_gn_f16, g_idx):
    """AutoGPTQ with ``desc_act``: input channel k belongs to group ``g_idx[k]`` (any order, ``group_size`` channels
    per group) -- W[k, n] = (q[k, n] - z[g_idx[k], n]) * s[g_idx[k], n] in fp32 -> [K, N]  (QuantLinear.forward of
    auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py: ``zeros[g_idx]``, ``scales[g_idx]``)."""
    g = np.asarray(g_idx).astype(np.int64)
    return (q_kn.astype(np.float32) - zeros_gn.astype(np.float32)[g]) * scales_gn_f16.astype(np.float32)[g]
