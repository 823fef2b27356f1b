"""lite_llama_amd -- MI355X (gfx950) native decode hot path of lite_llama.

Hand-written HIP kernels behind a C ABI (include/lite_llama_amd.h), mirrored in Python with
the reference's kernel-layer signatures (lite_llama_amd.kernels).  There is no CPU path."""

__version__ = "0.1.0"
