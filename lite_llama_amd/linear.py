"""Linear layers: quant-method dispatch + Megatron column/row sharding -- mirror of
lite_llama/models/linear.py:31-177.  ``RowParallelLinear.forward = all_reduce_tp(apply_linear(x))``
is the block's single collective."""

from __future__ import annotations

import torch
import torch.nn as nn

from .distributed.parallel_state import all_reduce_tp, divide, get_tp_world_size
from .quantization import QuantConfig, get_linear_method


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, *, bias: bool = False, quant: QuantConfig | None = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.quant = quant
        self.quant_method = get_linear_method(quant)
        self.quant_method.create_weights(self, input_size, output_size)
        self.bias = (nn.Parameter(torch.empty(output_size, dtype=torch.float16), requires_grad=False) if bias else None)

    def apply_linear(self, x: torch.Tensor) -> torch.Tensor:
        return self.quant_method.apply(self, x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.apply_linear(x)

    @torch.no_grad()
    def quantize_(self, quant: QuantConfig) -> None:
        """Replace a loaded fp16 weight with its quantised form (already-quantised layers untouched)."""
        if self.quant is not None:
            return
        method = get_linear_method(quant)
        method.convert_from_fp16(self, quant)
        self.quant = quant
        self.quant_method = method


class ReplicatedLinear(LinearBase):
    pass


class ColumnParallelLinear(LinearBase):
    """Output features split across TP ranks; no communication."""

    def __init__(self, input_size, output_size, *, bias=False, quant=None, what="output features"):
        local_out = divide(output_size, get_tp_world_size(), what)
        _check_shard_alignment(quant, local_out, what)
        super().__init__(input_size, local_out, bias=bias, quant=quant)
        self.full_output_size = output_size


class RowParallelLinear(LinearBase):
    """Contracted features split across TP ranks; partial sums are all-reduced."""

    def __init__(self, input_size, output_size, *, bias=False, quant=None, what="input features"):
        if bias:
            raise ValueError("RowParallelLinear cannot carry a bias: it would be added once per rank")
        local_in = divide(input_size, get_tp_world_size(), what)
        _check_shard_alignment(quant, local_in, what)
        super().__init__(local_in, output_size, quant=quant)
        self.full_input_size = input_size

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return all_reduce_tp(self.apply_linear(x))


def _check_shard_alignment(quant, local_size: int, what: str) -> None:
    if quant is not None and not quant.shard_is_aligned(local_size):
        raise ValueError(
            f"tensor-parallel shard of {what} is {local_size} channels, which is not a "
            f"multiple of the {quant.format} scale block ({quant.group_n}x{quant.group_k}); "
            "use a smaller tensor_parallel_size"
        )
