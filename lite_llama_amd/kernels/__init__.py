"""The kernel layer: the 16 names lite_llama/kernels/__init__.py:23-39 exports, same
signatures, backed by hand-written HIP for gfx950 behind the C ABI (include/lite_llama_amd.h)."""

from .activations import gelu, leaky_relu, relu, tanh
from .attention import flash_attention2_no_pad, flash_decoding, flash_decoding_fp8kv
from .fused_moe import fused_moe, moe_align_block_size
from .kv_cache import update_kv_buffer, update_kv_buffer_fp8, update_kv_index
from .norm_act import rope_emb_forward, skip_rmsnorm, swiglu_forward
from .quantization import smoothquant_matmul, w4a16_matmul, w8a16_matmul

__all__ = [
    "flash_attention2_no_pad",
    "flash_decoding",
    "fused_moe",
    "gelu",
    "leaky_relu",
    "moe_align_block_size",
    "relu",
    "rope_emb_forward",
    "skip_rmsnorm",
    "swiglu_forward",
    "tanh",
    "update_kv_buffer",
    "update_kv_index",
    "w4a16_matmul",
    "w8a16_matmul",
    "smoothquant_matmul",
]
