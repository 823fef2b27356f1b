"""TEST INFRASTRUCTURE ONLY -- the fp8 (OCP e4m3fn) KV cache of csrc/kv_ops.hip::ll_update_kv_buffer_fp8 and
csrc/flash_decoding.hip::ll_flash_decoding_fp8kv, restated with torch on the CPU.

PARITY UNPINNED against the reference: its pool is fp16 (lite_llama/executor/kv_cache_manager.py:197-216), an fp8 pool
is SURVEY 8f-3's extension.  What pins the device code: the quantiser must equal ``torch.float8_e4m3fn`` conversion
(IEEE-style round to nearest even, clamp to +-448) bit for bit, and attention over the fp8 pool must equal the pinned
fp16 oracle (oracle.flash_decoding, itself pinned by reference-generated fixtures) evaluated on the WIDENED pool."""

from __future__ import annotations

import torch

E4M3_MAX = 448.0


def quantize_rows(values: torch.Tensor, num_k_heads: int, k_scale: float, v_scale: float) -> torch.Tensor:
    """``[tokens, heads, hd]`` fp16 / bf16 -> uint8 e4m3fn codes; head h < num_k_heads uses k_scale, the rest v_scale."""
    x = values.float().clone()
    x[:, :num_k_heads] /= k_scale
    x[:, num_k_heads:] /= v_scale
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).view(torch.uint8)


def widen(codes: torch.Tensor, scale: float) -> torch.Tensor:
    """uint8 e4m3fn codes -> fp32 values * scale."""
    return codes.view(torch.float8_e4m3fn).float() * scale
