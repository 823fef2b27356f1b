"""ctypes binding of the C-ABI shared library (include/lite_llama_amd.h).

The product path has NO fallback: if the HIP library is missing or a symbol is absent
this module raises, loudly.  Build it with ``python -m lite_llama_amd.build``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LL_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "liblite_llama_amd.so")  # override: A/B builds

LL_F16, LL_BF16, LL_F32 = 0, 1, 2
LL_I32, LL_I64 = 0, 1
LL_W_F16, LL_W_FP8E4M3, LL_W_INT8 = 0, 1, 2
ABI_VERSION = 1

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes, exactly the declarations of include/lite_llama_amd.h
SIGNATURES = {
    "ll_abi_version": [],
    "ll_skip_rmsnorm": [P, P, P, P, L, L, F, I, P],
    "ll_swiglu": [P, P, P, L, L, I, P],
    "ll_activation": [P, P, L, I, I, P],
    "ll_rope": [P, P, P, P, L, I, I, I, L, L, L, L, L, L, L, I, I, P],
    "ll_rope_kv_update": [P, P, P, P, P, P, L, I, I, I, L, L, L, L, L, L, L, L, L, I, I, I, P, P],
    "ll_update_kv_buffer": [P, P, P, L, I, I, L, L, L, L, I, P],
    "ll_update_kv_index": [P, P, P, P, L, L, L, I, I, I, P],
    "ll_flash_decoding_num_partitions": [L],
    "ll_flash_decoding_group_waves": [L, L],
    "ll_flash_decoding": [P, P, P, P, P, P, P, P, P, I, I, I, I, L, F, L, L, L, L, L, L, L, L, L, I, I, I, P, P],
    "ll_decode_attention": [P, P, P, L, P, P, L, P, P, I, P, P, P, P, P, P, P, I, I, I, I, L, F, L, L, L, L, L, L, L, L, L,
                            I, I, I, P, P, P, F, P],
    "ll_decode_attention_partials": [P, P, I, P, P, P, L, P, P, I, P, P, P, P, P, I, I, I, I, L, F, L, L, L, L, L, L, L, I, I, I,
                                     P, P, F, P, P, P],
    "ll_flash_attention_nopad": [P, P, P, P, P, P, I, I, I, I, L, F, L, L, L, L, L, L, L, L, I, I, I, P],
    "ll_gemm_workspace": [L, L, L, P, P],
    "ll_w4a16_matmul": [P, P, P, P, P, P, L, L, L, I, L, L, L, P, P, P],
    "ll_w4a16_pack_scales": [P, P, P, L, L, L, P],
    "ll_w4a16_pack_weights": [P, P, L, L, L, P],
    "ll_w4a16_unpack_weights": [P, P, L, L, L, P],
    "ll_w4a16_prepacked_supported": [L, L, L, I],
    "ll_w4a16_partials_count": [L, L, L, I],
    "ll_w4a16_partials_count_ex": [L, L, L, I, I],
    "ll_w4a16_short_plan": [L, L, L, I, P],
    "ll_w4a16_short_full_plan": [L, L, L, I, P],
    "ll_skip_rmsnorm_partials": [P, P, I, P, P, L, L, F, I, P],
    "ll_skip_rmsnorm_slots": [P, P, I, P, P, L, L, F, I, P],
    "ll_w4a16_matmul_prepacked": [P, P, P, P, P, L, L, L, I, L, P, P, I, P],
    "ll_w4a16_v4_plan": [L, L, L, I, I, P],
    "ll_w4a16_mtiled_supported": [L, L, L, I],
    "ll_w4a16_matmul_prepacked_mtiled": [P, P, P, P, P, L, L, L, I, L, I, P],
    "ll_w4a16_v3_plan": [L, L, L, I, I, P],
    "ll_w8a16_matmul": [P, P, P, P, P, L, L, L, I, L, I, L, L, L, L, P, P, P],
    "ll_quantize_activations_int8": [P, P, P, L, L, L, P],
    "ll_dense16_matmul": [P, P, P, P, L, L, L, L, L, I, P, P],
    "ll_w8a8_rows_supported": [L, L, L, I],
    "ll_w8a8_rows_matmul": [P, P, P, P, P, P, L, L, L, L, L, I, P],
    "ll_dense16_rows_supported": [L, L, L, I],
    "ll_dense16_rows_matmul": [P, P, P, P, L, L, L, L, L, I, I, P],
    "ll_dense_partials_count": [L, L, L, I, I],
    "ll_dense_partials": [P, P, P, P, L, L, L, I, L, I, L, L, L, L, I, P],
    "ll_skip_rmsnorm_q8": [P, P, P, P, P, I, P, P, P, P, P, L, L, F, P],
    "ll_w8a8_finish_swiglu": [P, P, I, P, P, L, L, P],
    "ll_quant_act_cached_try": [P, P, P, L, L, L, P],
    "ll_w8a8_matmul": [P, P, P, P, P, P, L, L, L, L, P, P, P, P],
    "ll_w8_mtiled_supported": [L, L, L, I, L],
    "ll_moe_align_block_size": [P, I, L, I, I, P, P, P, P],
    "ll_moe_align_workspace_ints": [L, I],
    "ll_moe_align_block_size_ws": [P, I, L, I, I, P, P, P, P, L, P],
    "ll_moe_gemm": [P, P, P, P, P, P, P, P, L, L, I, L, L, I, I, I, I, L, L, L, L, L, L, L, I, P],
    "ll_silu_and_mul": [P, P, L, L, I, P],
    "ll_silu_and_mul_pairs": [P, P, L, L, I, P],
    "ll_moe_sum": [P, P, L, I, L, I, P],
    "ll_moe_route_topk": [P, P, P, L, I, L, I, I, I, P],
    "ll_moe_router_supported": [L, I, L],
    "ll_moe_router_workspace_floats": [L, I, L],
    "ll_moe_router": [P, P, P, P, P, L, I, L, L, L, I, I, I, I, P, P, P, P, P],
    "ll_argmax": [P, P, L, L, L, I, P],
    "ll_decode_advance": [P, L, P, P, P, P, P, P, P, P, L, L, I, P],
    "ll_slot_advance": [P, P, P, P, P, P, P, L, L, I, I, P],
    "ll_kv_alloc_scratch_bytes": [L],
    "ll_kv_alloc": [P, L, L, I, P, P, P, P, P],
    "ll_kv_ref_update": [P, L, P, L, I, I, P, P],
    "ll_update_kv_buffer_fp8": [P, P, P, L, I, I, I, L, L, L, L, F, F, I, I, P],
    "ll_flash_decoding_fp8kv": [P, P, P, P, P, P, P, P, P, I, I, I, I, L, F, F, F, L, L, L, L, L, L, L, L, L, I, I, P, P],
    "ll_tp_shared_alloc": [P, L],
    "ll_tp_shared_free": [P],
    "ll_tp_ipc_export": [P, P],
    "ll_tp_ipc_open": [P, P],
    "ll_tp_ipc_close": [P],
    "ll_tp_oneshot_flag_words": [I, I],
    "ll_tp_error_word": [P, L, P],
    "ll_tp_allreduce_oneshot": [P, L, I, P, P, I, I, L, I, P, P],
    "ll_tp_allreduce_norm_partials": [P, P, I, P, P, L, L, F, I, P, P, I, I, L, I, P, P],
    "ll_kv_paged_reset": [P, P, P, L, L, P],
    "ll_kv_paged_extend": [P, P, P, L, P, P, P, I, L, I, L, I, P, L, P, L, P],
    "ll_kv_paged_release": [P, P, P, L, P, P, L, P],
    "ll_w4_from_awq": [P, P, P, P, P, P, L, L, L, P],
    "ll_w4_from_gptq": [P, P, P, P, P, P, L, L, L, I, P],
    "ll_repetition_penalty": [P, P, P, P, P, F, L, L, L, L, L, L, L, I, I, P],
    "ll_sample_top_p": [P, P, P, P, P, P, L, L, L, I, P],
    "ll_argmax_split": [P, P, L, L, L, I, P, I, P],
}

_RETURNS_I64 = {"ll_kv_alloc_scratch_bytes", "ll_tp_oneshot_flag_words", "ll_moe_router_workspace_floats", "ll_moe_align_workspace_ints"}
_lib = None


class KernelError(RuntimeError):
    """A C-ABI entry point returned a negative status."""


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"lite_llama_amd: HIP library not found at {LIB_PATH}; run "
                "`python -m lite_llama_amd.build` (there is no CPU/eager fallback)"
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            if os.environ.get("LL_LIB_OVERRIDE") and not hasattr(handle, name):
                continue  # A/B builds of an older source (benchmarks only): a call to the missing entry still fails loudly
            fn = getattr(handle, name)  # AttributeError if the symbol is missing -> loud
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int64 if name in _RETURNS_I64 else c_int
        if handle.ll_abi_version() != ABI_VERSION:
            raise RuntimeError("lite_llama_amd: ABI version mismatch between _lib.py and the .so")
        _lib = handle
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return LL_F16
    if dt == torch.bfloat16:
        return LL_BF16
    if dt == torch.float32:
        return LL_F32
    raise TypeError(f"unsupported dtype {dt}")


def index_width(t: torch.Tensor) -> int:
    if t.dtype == torch.int32:
        return LL_I32
    if t.dtype == torch.int64:
        return LL_I64
    raise TypeError(f"index tensors must be int32 or int64, got {t.dtype}")


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "lite_llama_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback"
            )


def check(status: int, what: str) -> None:
    if status != 0:
        names = {-1: "LL_ERR_DTYPE", -2: "LL_ERR_SHAPE", -3: "LL_ERR_ARG", -4: "LL_ERR_LAUNCH"}
        raise KernelError(f"{what} failed: {names.get(status, status)}")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


# --------------------------------------------------------------------------- #
# Scratch the kernels WRITE (split-K slabs / merge counters, flash-decoding merge counters): one set per
# (device, stream) for eager calls and one per device for graph captures, grown on demand OUTSIDE capture
# (kernels never allocate; counters are zero between calls).  Two eager streams never share a set, and an
# eager call never shares one with a replaying graph.  Graphs captured through this module share the
# device's capture set: replay them on one stream at a time (what DecodeEngine / SlotRunner do).
# --------------------------------------------------------------------------- #
_gemm_ws: dict = {}
_scratch_keepalive: list = []  # captured hipGraphs may still point at outgrown buffers


def _device_index(device) -> int:
    """``torch.device("cuda")`` carries no index; tensors' devices always do -- keys use the resolved one."""
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def scratch_keys(device: torch.device):
    """(key to use now, key of the capture set)."""
    device = torch.device(device)
    idx = _device_index(device)
    cap = (device.type, idx, "capture")
    if torch.cuda.is_current_stream_capturing():
        return cap, cap
    return (device.type, idx, torch.cuda.current_stream(idx).cuda_stream), cap


def _grow_gemm_ws(key, device, floats: int, ints: int):
    ws = _gemm_ws.get(key)
    if ws is None or ws[0].numel() < floats or ws[1].numel() < ints:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(
                "GEMM split-K workspace must be sized before graph capture (run one eager warm-up step)"
            )
        nf = max(floats, ws[0].numel() if ws else 0)
        ni = max(ints, ws[1].numel() if ws else 0, 4096)
        ws = (
            torch.empty(nf, dtype=torch.float32, device=device),
            torch.zeros(ni, dtype=torch.int32, device=device),
        )
        _gemm_ws[key] = ws
        _scratch_keepalive.append(ws)
    return ws


def gemm_scratch_error(device: torch.device) -> bool:
    """True if a merge-counter buffer of ``device`` that the CURRENT stream's work uses -- this stream's eager set and the
    device's capture set (graphs are replayed on one stream at a time) -- holds a non-zero word after the stream has been
    synchronised: the buffers are all zero at rest by construction, so a non-zero word is the sticky trace of a GEMM merge
    that gave up on a contributor (its tile was written as NaN) -- csrc/gemm_w4_v3.hip.  Other eager streams' sets are not
    looked at: their counters are legitimately non-zero in the middle of a GEMM."""
    key, cap = scratch_keys(device)
    torch.cuda.current_stream(key[1]).synchronize()
    bad = False
    for k in {key, cap}:
        ws = _gemm_ws.get(k)
        if ws is not None and ws[1].numel():
            # a few KB copied to the host and inspected there: no reduction kernel (whose first use in a process loads a code
            # object -- 5 - 22 ms measured at the end of a 20-step bench run)
            bad = bad or bool(ws[1].cpu().numpy().any())
    return bad


def gemm_workspace(device: torch.device, m: int, n: int, k: int):
    floats, ints = c_int64(0), c_int64(0)
    lib().ll_gemm_workspace(m, n, k, ctypes.byref(floats), ctypes.byref(ints))
    key, cap = scratch_keys(device)
    if key != cap:  # every shape warmed up eagerly is capturable afterwards
        _grow_gemm_ws(cap, device, floats.value, ints.value)
    return _grow_gemm_ws(key, device, floats.value, ints.value)
