"""Tensor-parallel shard plans BEYOND the reference's divisibility rules (SURVEY 8e "extension").

The reference cuts every sharded dimension into ``tp`` equal parts and refuses anything else: ``tp | Hq``, ``tp | Hkv``
(models/base.py:169-171 -> distributed/parallel_state.py:194-202) and shard sizes that are multiples of the scale block
(models/linear.py:164-177, quantization/config.py:244-251).  Qwen2.5-7B (28 query / 4 key-value heads, intermediate
18944 = 148 int4 groups of 128) therefore stops at TP = 4.  A plan generalises the cut without touching the kernels:

* key/value heads: for ``tp > Hkv`` (``Hkv | tp``) every KV head is REPLICATED on ``r = tp / Hkv`` ranks, which share its
  ``G = Hq / Hkv`` query heads in contiguous parts as even as possible (7 -> 4 + 3).  A rank's attention is an ordinary
  grouped-query attention over its own query heads and ONE KV head; its o_proj shard is its heads' columns.  The KV cache
  per rank stays at one head (it cannot shrink below that).
* MLP intermediate: whole quantisation groups per rank, as even as possible (148 groups over 8 ranks = 4 x 19 + 4 x 18):
  no shard cuts a scale group, the int4 engines see multiples of 128 in both the column- and the row-parallel
  projection, at the price of a 2.7 % load imbalance.  (The alternative named in SURVEY 8e -- equal 2368-wide shards with
  the cut groups re-grouped to 64 at load time -- would take the projections off the 128-group decode engine.)

When the reference's rules hold, a plan reproduces its equal cuts exactly (``uniform`` is True) and nothing changes.
Results are unchanged by construction: a row-parallel projection sums over disjoint contracted ranges whatever their
sizes, and a column-parallel one concatenates output ranges.
"""

from __future__ import annotations

from dataclasses import dataclass


def split_even(total_units: int, parts: int) -> list[tuple[int, int]]:
    """``parts`` contiguous (start, count) ranges over ``total_units``; the first ``total_units % parts`` get one more."""
    if parts < 1 or total_units < parts:
        raise ValueError(f"cannot split {total_units} units into {parts} non-empty parts")
    base, rem = divmod(total_units, parts)
    out, start = [], 0
    for j in range(parts):
        n = base + (1 if j < rem else 0)
        out.append((start, n))
        start += n
    return out


@dataclass(frozen=True)
class ShardPlan:
    tp: int
    head_dim: int
    q_heads: tuple      # per rank (first query head, count)
    kv_heads: tuple     # per rank (first KV head, count); ranks that share a head list the same range
    inter: tuple        # per rank (first MLP intermediate channel, count)
    uniform: bool       # True: exactly the reference's equal cuts

    def q_range(self, rank):
        s, n = self.q_heads[rank]
        return s * self.head_dim, n * self.head_dim

    def kv_range(self, rank):
        s, n = self.kv_heads[rank]
        return s * self.head_dim, n * self.head_dim

    def inter_range(self, rank):
        return self.inter[rank]

    def describe(self) -> str:
        if self.uniform:
            return f"tp{self.tp}: equal cuts (the reference's rule)"
        qs = sorted({n for _, n in self.q_heads})
        rep = self.tp // len({s for s, _ in self.kv_heads}) if self.kv_heads else 1
        its = sorted({n for _, n in self.inter})
        return (f"tp{self.tp} extension: {'/'.join(map(str, qs))} query heads per rank, every KV head on {rep} rank(s), "
                f"MLP intermediate {'/'.join(map(str, its))} channels per rank (whole scale groups)")


def scale_unit(quant, intermediate: int) -> int:
    """Width the MLP-intermediate cut must respect for a run's quantisation: the scale-group width of a group format
    (int4 / int8 groups, fp8 blocks), 1 for unquantised and per-channel weights -- there the reference's equal cut
    ``intermediate / tp`` applies whenever it divides (models/weights.py:125-134), whatever 128 says (ADVICE round 3:
    11008 channels at TP = 8 are 1376 per rank, not an 11 / 10-group extension plan)."""
    if quant is None:
        return 1
    gk = int(getattr(quant, "group_k", 0) or 0)
    return gk if 1 < gk < intermediate else 1


def make_plan(num_heads: int, num_kv_heads: int, head_dim: int, intermediate: int, tp: int, unit: int = 128) -> ShardPlan:
    """``unit``: the scale-group width the intermediate cut must respect (int4 group / fp8 block: 128; 1 for unquantised
    or per-channel formats -- the decode engines still prefer 128)."""
    if tp < 1:
        raise ValueError("tp must be >= 1")
    uniform = True
    # ---- attention heads ----
    if num_kv_heads % tp == 0 and num_heads % tp == 0:
        hq, hk = num_heads // tp, num_kv_heads // tp
        q = [(r * hq, hq) for r in range(tp)]
        kv = [(r * hk, hk) for r in range(tp)]
    elif tp % num_kv_heads == 0 and num_heads % num_kv_heads == 0:
        uniform = False
        rep, group = tp // num_kv_heads, num_heads // num_kv_heads
        if group < rep:
            raise ValueError(f"{group} query heads per KV head cannot be shared by {rep} ranks")
        parts = split_even(group, rep)
        q = [(r // rep * group + parts[r % rep][0], parts[r % rep][1]) for r in range(tp)]
        kv = [(r // rep, 1) for r in range(tp)]
    else:
        raise ValueError(f"{num_heads} query / {num_kv_heads} key-value heads cannot be laid out on {tp} tensor-parallel ranks "
                         "(need tp | Hkv, or Hkv | tp with whole query heads per rank)")
    # ---- MLP intermediate ----
    if intermediate % tp == 0 and (intermediate // tp) % unit == 0:
        step = intermediate // tp
        inter = [(r * step, step) for r in range(tp)]
    else:
        if intermediate % unit != 0:
            raise ValueError(f"intermediate size {intermediate} is not a multiple of the scale group {unit}")
        uniform = False
        inter = [(s * unit, n * unit) for s, n in split_even(intermediate // unit, tp)]
    return ShardPlan(tp, head_dim, tuple(q), tuple(kv), tuple(inter), uniform)


def admissible_tp(num_heads: int, num_kv_heads: int, head_dim: int, intermediate: int, world: int, unit: int = 128) -> int:
    """Largest TP degree dividing ``world`` that has a plan (the remaining factor runs as data-parallel replicas)."""
    for cand in sorted({d for d in range(1, world + 1) if world % d == 0}, reverse=True):
        try:
            make_plan(num_heads, num_kv_heads, head_dim, intermediate, cand, unit)
            return cand
        except ValueError:
            continue
    return 1


# ------------------------------------------------------------------------------------------------------------------ #
# the plan in force for the model being built / loaded (None: the reference's equal cuts)
# ------------------------------------------------------------------------------------------------------------------ #
_ACTIVE: ShardPlan | None = None
_ACTIVE_FULL: dict = {}


def set_active_plan(plan: ShardPlan | None, num_heads: int = 0, num_kv_heads: int = 0, intermediate: int = 0) -> None:
    """Registered by ``CausalLM.__init__`` so that the checkpoint loader (weights.py) cuts incoming tensors the way the
    modules were sized.  ``None`` restores the reference rule."""
    global _ACTIVE, _ACTIVE_FULL
    _ACTIVE = plan
    _ACTIVE_FULL = {} if plan is None else {"q": num_heads * plan.head_dim, "kv": num_kv_heads * plan.head_dim,
                                            "inter": intermediate}


def active_plan() -> ShardPlan | None:
    return _ACTIVE


_KIND_OF = (("self_attn.q_proj", "q"), ("self_attn.kv_proj", "kv"), ("self_attn.k_proj", "kv"), ("self_attn.v_proj", "kv"),
            ("self_attn.o_proj", "q"), ("mlp.gate_proj", "inter"), ("mlp.up_proj", "inter"), ("mlp.down_proj", "inter"))


def plan_range(module_name: str, size_along_cut: int, rank: int):
    """(start, count) of ``rank``'s slice of a tensor whose cut dimension has ``size_along_cut`` entries -- the full
    logical size (weights, biases), or a coarser image of it (scale grids: one entry per group; packed int4 words: one
    per 8 channels).  ``None`` when no plan is active or the module is not one it covers."""
    if _ACTIVE is None:
        return None
    for suffix, kind in _KIND_OF:
        if module_name.endswith(suffix):
            full = _ACTIVE_FULL[kind]
            start, count = {"q": _ACTIVE.q_range, "kv": _ACTIVE.kv_range, "inter": _ACTIVE.inter_range}[kind](rank)
            if size_along_cut == full:
                return start, count
            if size_along_cut <= 0 or full % size_along_cut != 0:
                raise ValueError(f"{module_name}: a tensor with {size_along_cut} entries along the cut is no image of the "
                                 f"{full}-wide dimension")
            ratio = full // size_along_cut
            if start % ratio or count % ratio:
                raise ValueError(f"{module_name}: the rank's range [{start}, {start + count}) does not fall on the tensor's "
                                 f"granularity of {ratio} channels")
            return start // ratio, count // ratio
    return None
