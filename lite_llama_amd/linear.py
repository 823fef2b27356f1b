"""Linear layers: quant-method dispatch + Megatron column/row sharding -- mirror of
lite_llama/models/linear.py:31-177.  ``RowParallelLinear.forward = all_reduce_tp(apply_linear(x))``
is the block's single collective."""

from __future__ import annotations

import os

import torch
import torch.nn as nn

from .distributed.parallel_state import all_reduce_tp, collective_forced, divide, get_tp_world_size
from .quantization import QuantConfig, get_linear_method


def _export_reference_weight(module, state_dict, prefix, local_metadata):
    """state_dict post-hook of every linear: a compacted int4 layer's ``weight`` parameter aliases the decode engine's load-time
    layout (permuted words) -- the EXPORTED entry is the reference-format tensor, rebuilt transiently (bit-exact inverse
    permutation); the model itself is not touched, so captured graphs keep replaying over the packed storage.
    Covers ``model.state_dict()`` and any ``submodule.state_dict()`` alike."""
    if getattr(module, "_w4_compact", False) or getattr(module, "_w4_compact_member", None) is not None:
        key = prefix + "weight"
        if key in state_dict:
            state_dict[key] = module.quant_method.reference_weight(module).detach()


def _refuse_load_into_compact(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    """load_state_dict pre-hook: checkpoint rows copied into a parameter that aliases the load-time layout would be read as
    permuted words -- silently wrong.  ``CausalLM.load_state_dict`` expands first; a direct submodule load has to."""
    if (getattr(module, "_w4_compact", False) or getattr(module, "_w4_compact_member", None) is not None) \
            and (prefix + "weight") in state_dict:
        raise RuntimeError(f"{prefix}weight aliases the decode engine's load-time layout (compact_weights): call "
                           "model.expand_weights() before loading a checkpoint into it")


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, *, bias: bool = False, quant: QuantConfig | None = None):
        super().__init__()
        self.register_state_dict_post_hook(_export_reference_weight)
        self.register_load_state_dict_pre_hook(_refuse_load_into_compact)
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

    def __init__(self, input_size, output_size, *, bias=False, quant=None, what="output features", local_size=None):
        """``local_size`` (extension, distributed/partition.py): this rank's share of the output features when the cut is
        not the reference's equal division; ``None`` -> ``output_size / tp`` and its error behaviour."""
        local_out = divide(output_size, get_tp_world_size(), what) if local_size is None else int(local_size)
        _check_shard_alignment(quant, local_out, what)
        super().__init__(input_size, local_out, bias=bias, quant=quant)
        self.full_output_size = output_size


class RowParallelLinear(LinearBase):
    """Contracted features split across TP ranks; partial sums are all-reduced."""

    def __init__(self, input_size, output_size, *, bias=False, quant=None, what="input features", local_size=None):
        if bias:
            raise ValueError("RowParallelLinear cannot carry a bias: it would be added once per rank")
        local_in = divide(input_size, get_tp_world_size(), what) if local_size is None else int(local_size)
        _check_shard_alignment(quant, local_in, what)
        super().__init__(local_in, output_size, quant=quant)
        self.full_input_size = input_size

    def forward(self, x: torch.Tensor, partials_ok: bool = False):
        """``partials_ok`` (extension): the caller feeds the result to ``skip_rmsnorm_partials`` and accepts a
        :class:`PartialSums` -- only taken without tensor parallelism (the all-reduce needs the finished sums)."""
        if partials_ok and not collective_forced() and hasattr(self.quant_method, "apply_partials"):
            from .distributed.parallel_state import shard_simulated, simulated_collective
            if get_tp_world_size() == 1 or shard_simulated():
                out = self.quant_method.apply_partials(self, x)
                if out is not None:
                    if shard_simulated():
                        # one rank's shard measured alone (bench.py --shard-sim): the fused partials + all-reduce + norm launch of
                        # the TP step becomes the local reducing norm + a same-size local copy standing in for the collective
                        simulated_collective(out.parts.shape[1] * out.parts.shape[2], x.dtype, x.device)
                    return out
            else:
                # tensor parallelism (round 3): the partials go to ONE launch that also carries the block's collective and
                # the add-and-normalise -- available when the one-shot all-reduce is enabled and the payload fits it
                # (decided from shapes only: every rank takes the same route)
                from .distributed.parallel_state import fused_reduce_norm_available
                rows = x.numel() // x.shape[-1] if x.shape[-1] else 0
                if fused_reduce_norm_available(rows, self.output_size, x.dtype) and not os.environ.get("LL_TP_NO_FUSED_NORM"):
                    if self._tp_fused_agreed(rows):
                        out = self.quant_method.apply_partials(self, x)
                        if out is None:  # (cannot happen after the agreement below; never a silent divergence)
                            raise RuntimeError("row-parallel projection: this rank's shard cannot leave split-K partials "
                                               "although the group agreed on the fused launch")
                        out.tp_reduce = True
                        return out
        return all_reduce_tp(self.apply_linear(x))

    def _tp_fused_agreed(self, rows: int) -> bool:
        """The fused partials + all-reduce + norm launch is taken only if EVERY rank of the group can take it (ADVICE round
        3: under extension plans a rank's contracted size differs -- 19 vs 18 groups -- and the engine's split count for it
        may be 0 on one rank only; the odd rank raised while its peers spun).  Decided ONCE per (layer, row count) by a MIN
        all-reduce of the local answers -- ``LL_W4_NO_PARTIALS`` on any rank simply turns the route off for the group -- on
        the first eager call (every rank walks the same call sequence); a call inside a graph capture that finds no
        decision takes the unfused route, on every rank alike."""
        # rank-uniform part first (format, group size, row count, bias: the same answer on every rank) -- formats that can
        # never take the fused launch must not enter a host-synchronising collective at all
        if not self._tp_partials_ok(rows):
            return False
        cache = self.__dict__.setdefault("_tp_fused_cache", {})
        if rows in cache:
            return cache[rows]
        if torch.cuda.is_current_stream_capturing():
            return False
        from .distributed.parallel_state import all_reduce_min
        mine = 0
        if not os.environ.get("LL_W4_NO_PARTIALS"):
            from . import _lib as L
            mine = 1 if L.lib().ll_w4a16_partials_count(rows, self.output_size, self.input_size, self.quant.group_k) > 0 else 0
        cache[rows] = bool(all_reduce_min(mine))
        return cache[rows]

    def _tp_partials_ok(self, rows: int) -> bool:
        """Whether EVERY rank's shard of this projection can leave split-K partials -- from properties all ranks share
        (format, group size, decode-shaped batch, output width) plus this rank's contracted size, which is a multiple of
        128 on every rank under both shard rules (equal cuts respect the scale group; extension plans cut whole groups)."""
        q = self.quant
        return (q is not None and getattr(q, "format", None) == "int4" and q.group_k % 128 == 0 and 1 <= rows <= 64
                and self.output_size % 128 == 0 and self.input_size % 128 == 0 and self.bias is None)


def _check_shard_alignment(quant, local_size: int, what: str) -> None:
    if quant is not None and not quant.shard_is_aligned(local_size):
        raise ValueError(
            f"tensor-parallel shard of {what} is {local_size} channels, which is not a "
            f"multiple of the {quant.format} scale block ({quant.group_n}x{quant.group_k}); "
            "use a smaller tensor_parallel_size"
        )


class MergedColumnLinear:
    """Several column-parallel linears that read the SAME input, run as one GEMM launch.

    MI355X-first extension (the reference launches q/kv and gate/up separately): at batch 64 a
    projection GEMM is dominated by per-launch fixed costs, so ``[q | k | v]`` and ``[gate | up]``
    go out as single launches.  The member layers keep their parameter NAMES and shapes (so
    ``state_dict`` round-trips and the reference's loaders still apply); their tensors are re-pointed
    at row-block views of one merged storage, so there is no second copy of the weights.  Row
    blocks of every format concatenate along N (int4 words, per-channel / per-group / per-block
    scale grids, biases); when a scale grid would not line up the members simply run unmerged.
    Not an ``nn.Module``: it owns no parameters of its own.
    """

    def __init__(self, layers, interleave: bool = False):
        """``interleave``: for a (gate, up) pair whose quant method offers ``apply_gate_up_swiglu``,
        store the two row-INTERLEAVED (row 2j = gate_j, row 2j+1 = up_j; the members become the
        stride-2 row views) so that :meth:`swiglu` can run projection + activation as one launch."""
        self.layers = list(layers)
        self.interleave = interleave
        self._key = None
        self._holder = None

    def _snapshot(self):
        return tuple((n, t.data_ptr(), t._version) for l in self.layers for n, t in l._parameters.items()
                     if t is not None) + tuple((id(l.quant_method), id(getattr(l, "act_perm", None))) for l in self.layers)

    @torch.no_grad()
    def refresh(self) -> bool:
        """(Re)build the merged storage if a member's parameters were replaced; returns whether the
        merged path can be used.  Never allocates inside a graph capture."""
        key = self._snapshot()
        if key == self._key:
            return self._holder is not None
        first = self.layers[0]
        if not first.weight.is_cuda or torch.cuda.is_current_stream_capturing():
            return False
        if any(getattr(l, "_w4_compact_member", None) is not None for l in self.layers):
            # the members' `weight` parameters are views of the decode engine's layout, not reference rows: merging them again
            # would permute garbage.  Somebody replaced a parameter of a compacted model.
            raise RuntimeError("a parameter of a compacted merged projection was replaced: call model.expand_weights() before "
                               "loading or assigning weights (CausalLM.compact_weights keeps only the decode engine's layout)")
        for l in self.layers:  # a member compacted on its own holds permuted words: merge reference-format rows only
            if getattr(l, "_w4_compact", False):
                l.quant_method.expand(l)
        self._holder = None
        ok = all(type(l.quant_method) is type(first.quant_method) and l.quant == first.quant
                 and l.input_size == first.input_size for l in self.layers)
        # activation-ordered GPTQ members (weights.py): one launch needs ONE input order
        perms = [getattr(l, "act_perm", None) for l in self.layers]
        if any(p is not None for p in perms):
            ok = ok and all(p is not None and torch.equal(p, perms[0]) for p in perms)
        il = (self.interleave and ok and len(self.layers) == 2 and hasattr(first.quant_method, "apply_gate_up_swiglu")
              and self.layers[0].output_size == self.layers[1].output_size)
        holder = _MergedHolder()
        holder.interleaved = il
        holder.input_size = first.input_size
        holder.output_size = sum(l.output_size for l in self.layers)
        holder.quant, holder.quant_method = first.quant, first.quant_method
        holder.act_perm = getattr(first, "act_perm", None)
        merged = {}
        for name in first._parameters:
            ts = [l._parameters.get(name) for l in self.layers]
            if all(t is None for t in ts):
                merged[name] = None
                continue
            if any(t is None for t in ts) or len({(t.dtype, t.shape[1:]) for t in ts}) != 1:
                ok = False
                break
            if il:
                merged[name] = torch.stack([t.data for t in ts], dim=1).reshape(-1, *ts[0].shape[1:])
            else:
                merged[name] = torch.cat([t.data for t in ts], dim=0)
            if first.quant is not None and ("scale" in name or "zeros" in name):
                if tuple(merged[name].shape) != tuple(first.quant.scale_shape(holder.output_size, holder.input_size)):
                    ok = False
                    break
        if ok:
            for name, cat in merged.items():
                setattr(holder, name, cat)
                if cat is None:
                    continue
                off = 0
                for i, l in enumerate(self.layers):
                    rows = l._parameters[name].shape[0]
                    view = cat[i::2] if il else cat[off:off + rows]
                    l._parameters[name] = nn.Parameter(view, requires_grad=False)
                    off += rows
            self._holder = holder
        self._key = self._snapshot()
        return ok

    def _repoint_members(self, name: str) -> None:
        """Members' parameters are views of the holder's merged tensor: rebuild them after the tensor was replaced."""
        h = self._holder
        cat = getattr(h, name)
        off = 0
        for i, l in enumerate(self.layers):
            rows = l._parameters[name].shape[0]
            view = cat[i::2] if h.interleaved else cat[off:off + rows]
            l._parameters[name] = nn.Parameter(view, requires_grad=False)
            off += rows

    def compact(self) -> int:
        """One resident copy of the merged int4 weights (quantization/methods.py::W4A16LinearMethod.compact): the holder's
        ``weight`` aliases the load-time layout; the members' ``weight`` parameters keep their shapes (views of the same
        storage -- permuted words, not reference rows) and rebuild their reference rows on demand.  Bytes released."""
        if not self.refresh():  # (builds the merged storage if no forward has run yet)
            return 0
        h = self._holder
        if h is None or not hasattr(h.quant_method, "compact"):
            return 0
        freed = h.quant_method.compact(h)
        if freed:
            off = 0
            rows_of = []
            for i, l in enumerate(self.layers):
                rows = l._parameters["weight"].shape[0]
                rows_of.append(slice(i, None, 2) if h.interleaved else slice(off, off + rows))
                off += rows
            self._repoint_members("weight")
            for l, rows in zip(self.layers, rows_of):
                l._w4_compact_member = (h, rows)
                for attr in ("_w4_prepacked",):
                    if hasattr(l, attr):
                        delattr(l, attr)
            self._key = self._snapshot()
        return freed

    def expand(self) -> None:
        h = self._holder
        if h is None or not getattr(h, "_w4_compact", False):
            return
        h.quant_method.expand(h)
        self._repoint_members("weight")
        for l in self.layers:
            if hasattr(l, "_w4_compact_member"):
                delattr(l, "_w4_compact_member")
        self._key = self._snapshot()

    def invalidate(self) -> None:
        """A checkpoint was copied THROUGH the members' parameters (views of the merged storage: the storage itself is up
        to date, no version counter moved): drop the layouts derived from it."""
        if self._holder is not None:
            for attr in ("_w4_prepacked", "_w4_packed"):
                if hasattr(self._holder, attr):
                    delattr(self._holder, attr)

    def __call__(self, x: torch.Tensor):
        """-> one output view per member (column blocks of the merged ``[..., sum N]`` result)."""
        out = self._holder.quant_method.apply(self._holder, x)
        if self._holder.interleaved:
            return out[..., 0::2], out[..., 1::2]
        return torch.split(out, [l.output_size for l in self.layers], dim=-1)

    def partials(self, x: torch.Tensor):
        """The merged projection left as fp32 split-K partials (:class:`PartialSums`, bias NOT added -- returned next
        to it) for a consumer that adds them up (``decode_attention_partials``); ``None`` when not served."""
        h = self._holder
        # (column-parallel: the planes and their consumer are this rank's own -- no collective is involved, so a TP shard takes the
        # route like the unsharded model; until round 6 it was declined under TP and a 1-MB q|k|v shard ran the finishing unit
        # loop on 30 workgroups: 11 us per layer at TP 8)
        if h is None or h.interleaved or not hasattr(h.quant_method, "apply_partials"):
            return None
        parts = h.quant_method.apply_partials(h, x, allow_bias=True, max_splits=8)  # the attention kernel adds <= 8 planes
        return None if parts is None else (parts, h.bias)

    def swiglu(self, x: torch.Tensor) -> torch.Tensor:
        """``silu(gate(x)) * up(x)`` for a (gate, up) pair: one launch when interleaved and the shape is
        in the decode engine, else the merged GEMM followed by ``swiglu_forward``."""
        from .kernels import swiglu_forward

        h = self._holder
        if h.interleaved:
            y = h.quant_method.apply_gate_up_swiglu(h, x)
            if y is not None:
                return y
        gate, up = self(x)
        return swiglu_forward(gate, up)


class _MergedHolder:
    """Attribute bag with the fields a quant method's ``apply`` reads from a layer."""
    bias = None
    act_perm = None
