"""Quantised-weight layout descriptor -- host-side mirror of the reference's ``QuantConfig``
(lite_llama/models/quantization/config.py:70-251): every scheme is a low-bit weight plus one
scale per ``group_n x group_k`` block.  ``quantization_config`` of a checkpoint's config.json is read by
``lite_llama_amd/weights.py::quantization_from_hf_config`` (config.py:99-157)."""

from __future__ import annotations

from dataclasses import dataclass

import torch

FP8 = "fp8"
INT8 = "int8"
INT4 = "int4"
SMOOTHQUANT = "smoothquant"
PER_CHANNEL_K = 1 << 30  # "covers K" sentinel used by the reference (config.py:161)

RUNTIME_SCHEMES = {"int8": INT8, "int8-blockwise": INT8, "fp8": FP8, "int4": INT4, "smoothquant": SMOOTHQUANT}


@dataclass(frozen=True)
class QuantConfig:
    format: str
    group_n: int
    group_k: int
    ignored: tuple = ()
    is_dynamic: bool = False

    # factories (config.py:159-208)
    @classmethod
    def fp8_block(cls, gn: int = 128, gk: int = 128) -> "QuantConfig":
        return cls(FP8, gn, gk)

    @classmethod
    def fp8_per_channel(cls) -> "QuantConfig":
        return cls(FP8, 1, PER_CHANNEL_K)

    @classmethod
    def int8_per_channel(cls) -> "QuantConfig":
        return cls(INT8, 1, PER_CHANNEL_K)

    @classmethod
    def int8_groupwise(cls, group_size: int = 128) -> "QuantConfig":
        return cls(INT8, 1, group_size)

    @classmethod
    def int4_groupwise(cls, group_size: int = 128) -> "QuantConfig":
        return cls(INT4, 1, group_size)

    @classmethod
    def smoothquant_per_channel(cls) -> "QuantConfig":
        return cls(SMOOTHQUANT, 1, PER_CHANNEL_K, is_dynamic=True)

    @classmethod
    def for_runtime_scheme(cls, name: str) -> "QuantConfig":
        fmt = RUNTIME_SCHEMES.get(name.lower())
        if fmt is None:
            raise ValueError(f"unknown runtime quantisation {name!r}; supported: {sorted(RUNTIME_SCHEMES)}")
        if fmt == INT8:
            return cls.int8_groupwise() if name.lower() == "int8-blockwise" else cls.int8_per_channel()
        if fmt == FP8:
            return cls.fp8_per_channel()
        if fmt == INT4:
            return cls.int4_groupwise()
        return cls.smoothquant_per_channel()

    # layout (config.py:211-251)
    @property
    def storage_dtype(self) -> torch.dtype:
        if self.format == FP8:
            return torch.uint8
        if self.format == INT4:
            return torch.int32
        return torch.int8

    @property
    def is_fp8(self) -> bool:
        return self.format == FP8

    @property
    def is_int4(self) -> bool:
        return self.format == INT4

    def scale_shape(self, out_features: int, in_features: int) -> tuple:
        if self.format == INT4:
            return (out_features, (in_features + self.group_k - 1) // self.group_k)
        return ((out_features + self.group_n - 1) // self.group_n,
                (in_features + self.group_k - 1) // self.group_k)

    def quantizes(self, module_name: str) -> bool:
        return not any(module_name == ig or module_name.startswith(ig + ".") for ig in self.ignored)

    def shard_is_aligned(self, size: int) -> bool:
        """Whether a TP shard of ``size`` channels keeps whole scale blocks (config.py:244-251)."""
        if self.group_n <= 1 and self.group_k >= PER_CHANNEL_K:
            return True
        if self.format == INT4:
            return size % self.group_k == 0
        return size % max(self.group_n, self.group_k) == 0
