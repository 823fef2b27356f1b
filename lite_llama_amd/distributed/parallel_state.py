"""Rank grid + the tensor-parallel collectives of the hot path -- mirror of
lite_llama/distributed/parallel_state.py:44-230.

One process per GPU; ``global_rank = dp_rank * tp_size + tp_rank`` (TP ranks contiguous); the
only data-path collective is the in-place SUM all-reduce of the ``[tokens, hidden]`` partial
sums after every row-parallel projection (2 per decoder layer).  Backend ``"nccl"`` IS RCCL on
ROCm (ring/tree over xGMI); on a CPU-only host (tests) the same code runs on ``gloo``.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

_TP_RANK = 0
_TP_WORLD_SIZE = 1
_TP_GROUP = None
_DP_RANK = 0
_DP_WORLD_SIZE = 1
_OWNS_PG = False
_FORCE_COLLECTIVE = False  # test knob (LL_TP_FORCE_COLLECTIVE): a world of ONE still issues its all-reduces on the backend
_ONESHOT = None            # peer-mapped one-shot all-reduce state (enable_oneshot_all_reduce), or None: the backend's collective
_SIMULATED = False         # simulate_shard(): ONE rank's shard on one device, no peers (bench.py --shard-sim)
_SIM_SCRATCH: dict = {}


def grid_coordinates(global_rank: int, tp_size: int, dp_size: int) -> tuple[int, int]:
    """``global_rank -> (dp_rank, tp_rank)`` for the layout ``dp_rank * tp_size + tp_rank``."""
    if tp_size < 1 or dp_size < 1:
        raise ValueError(f"tp_size and dp_size must be >= 1, got {tp_size} and {dp_size}")
    world = tp_size * dp_size
    if not 0 <= global_rank < world:
        raise ValueError(f"global_rank {global_rank} is outside a {dp_size}x{tp_size} grid of {world} ranks")
    return global_rank // tp_size, global_rank % tp_size


def _backend() -> str:
    # LL_DIST_BACKEND=gloo: test knob -- several ranks on ONE GPU (RCCL refuses that), collectives staged
    # through the host; the sharded kernels still run on the device
    forced = os.environ.get("LL_DIST_BACKEND")
    if forced:
        return forced
    return "nccl" if torch.cuda.is_available() else "gloo"


def init_parallel(global_rank: int = 0, tp_size: int = 1, dp_size: int = 1, master_port: int = 29500) -> None:
    """Place this process in the ``dp_size x tp_size`` grid; creates one TP group per replica.
    Blocks in the rendezvous until all ranks joined when ``tp_size > 1``."""
    global _TP_RANK, _TP_WORLD_SIZE, _TP_GROUP, _DP_RANK, _DP_WORLD_SIZE, _OWNS_PG, _FORCE_COLLECTIVE
    _DP_RANK, _TP_RANK = grid_coordinates(global_rank, tp_size, dp_size)
    _DP_WORLD_SIZE = dp_size
    _TP_WORLD_SIZE = tp_size
    # LL_TP_FORCE_COLLECTIVE=1 (tests): a single rank joins a real process group and every row-parallel projection
    # really calls the backend's all-reduce -- proves RCCL collectives inside a captured step on a one-GPU box
    _FORCE_COLLECTIVE = bool(os.environ.get("LL_TP_FORCE_COLLECTIVE")) and tp_size * dp_size == 1
    if tp_size <= 1 and not _FORCE_COLLECTIVE:
        return
    world = tp_size * dp_size
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(master_port))
    if not dist.is_initialized():
        kw = {}
        if torch.cuda.is_available() and _backend() == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=_backend(), rank=global_rank, world_size=world, **kw)
        _OWNS_PG = True
    # every rank creates every group, in the same order (new_group is itself a collective)
    for replica in range(dp_size):
        members = list(range(replica * tp_size, (replica + 1) * tp_size))
        group = dist.new_group(members, backend=_backend())
        if replica == _DP_RANK:
            _TP_GROUP = group


def init_tensor_parallel(rank: int = 0, world_size: int = 1, master_port: int = 29500) -> None:
    init_parallel(global_rank=rank, tp_size=world_size, dp_size=1, master_port=master_port)


def destroy_parallel() -> None:
    global _TP_RANK, _TP_WORLD_SIZE, _TP_GROUP, _DP_RANK, _DP_WORLD_SIZE, _OWNS_PG, _ONESHOT
    if _ONESHOT is not None:
        _ONESHOT.close()
        _ONESHOT = None
    if _TP_GROUP is not None and _OWNS_PG and dist.is_initialized():
        dist.destroy_process_group()
    _TP_RANK, _TP_WORLD_SIZE, _TP_GROUP, _DP_RANK, _DP_WORLD_SIZE, _OWNS_PG = 0, 1, None, 0, 1, False
    global _FORCE_COLLECTIVE, _SIMULATED
    _FORCE_COLLECTIVE = False
    _SIMULATED = False
    _SIM_SCRATCH.clear()


def collective_forced() -> bool:
    return _FORCE_COLLECTIVE


# ------------------------------------------------------------------------------------- #
# one-shot all-reduce over peer-mapped buffers (SURVEY 5 / 8e native target; csrc/tp_allreduce.hip)
# ------------------------------------------------------------------------------------- #
class _OneShot:
    """This rank's staging buffer + flag words (fine-grained device memory), the peers' mappings of theirs, and the
    launch counter.  Built once per TP group; every later all-reduce of a fitting 16-bit tensor is ONE kernel launch with
    no host work (capturable in the decode graph)."""

    def __init__(self, max_elems: int, blocks: int):
        """Every rank walks the same sequence of collectives whatever happens locally (a rank that failed to allocate or
        to map still takes part in the exchanges and the verdict), so a partial failure ends in a common RuntimeError,
        never in a rank waiting for a peer that left."""
        import ctypes

        from .. import _lib as L

        lib = L.lib()
        self.lib, self.L = lib, L
        self.world, self.rank = _TP_WORLD_SIZE, _TP_RANK
        self.stage_elems = (max_elems + 7) // 8 * 8
        self.blocks = blocks
        self.flag_words = int(lib.ll_tp_oneshot_flag_words(blocks, self.world))
        self._mine, self._opened = [], []
        problem = None
        handles = []
        try:
            for nbytes in (2 * self.stage_elems * 2, self.flag_words * 4):
                ptr = ctypes.c_void_p()
                L.check(lib.ll_tp_shared_alloc(ctypes.byref(ptr), nbytes), "tp_shared_alloc")
                self._mine.append(ptr)
            for ptr in self._mine:
                buf = ctypes.create_string_buffer(64)
                L.check(lib.ll_tp_ipc_export(ptr, buf), "tp_ipc_export")
                handles.append(buf.raw)
        except Exception as exc:
            problem = f"rank {self.rank}: {type(exc).__name__}: {exc}"
        gathered = [None] * self.world
        me = torch.cuda.current_device()
        ident = (me, str(getattr(torch.cuda.get_device_properties(me), "uuid", "")) or None)
        dist.all_gather_object(gathered, (self.rank, handles if problem is None else None, problem, ident), group=_TP_GROUP)
        # which device every rank's staging buffer lives on: (index, uuid) -- distinct entries = the peer mappings cross
        # devices (xGMI / PCIe peer access), equal entries = ranks sharing one device (the debugging set-up of the tests)
        self.peer_devices = [g[3] for g in gathered]
        failures = [g[2] for g in gathered if g[2]]
        stage, flags = [None] * self.world, [None] * self.world
        if not failures:
            try:
                for r, hs, _, _ in gathered:
                    if r == self.rank:
                        stage[r], flags[r] = self._mine[0].value, self._mine[1].value
                        continue
                    ptrs = []
                    for h in hs:
                        out = ctypes.c_void_p()
                        L.check(lib.ll_tp_ipc_open(ctypes.create_string_buffer(h, 64), ctypes.byref(out)), "tp_ipc_open")
                        ptrs.append(out)
                        self._opened.append(out)
                    stage[r], flags[r] = ptrs[0].value, ptrs[1].value
            except Exception as exc:
                problem = f"rank {self.rank}: {type(exc).__name__}: {exc}"
        # the verdict doubles as the barrier: nobody launches before everybody has mapped everybody
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, problem, group=_TP_GROUP)
        failures += [v for v in verdicts if v]
        if failures:
            self.close()
            raise RuntimeError("one-shot all-reduce set-up failed: " + "; ".join(sorted(set(failures))))
        self.stage_arr = (ctypes.c_void_p * self.world)(*stage)
        self.flag_arr = (ctypes.c_void_p * self.world)(*flags)
        self.epoch_done = torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))

    def crosses_devices(self) -> bool:
        """True when no two ranks of the group stage on the same device (the real multi-GPU set-up)."""
        return len(set(self.peer_devices)) == self.world

    def fits(self, t: torch.Tensor) -> bool:
        """Decided from what is IDENTICAL on every rank of the group (dtype, element count) -- never from this rank's
        storage (alignment, strides): ranks taking different collectives for one call would hang."""
        return (t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and t.numel() % 8 == 0
                and 0 < t.numel() <= self.stage_elems)

    def all_reduce(self, t: torch.Tensor) -> None:
        import ctypes

        L = self.L
        if not t.is_contiguous() or t.data_ptr() % 16 != 0:  # this rank's storage is unusual: same collective, via a copy
            tmp = t.contiguous().clone()
            self.all_reduce(tmp)
            t.copy_(tmp)
            return
        L.check(self.lib.ll_tp_allreduce_oneshot(
            t.data_ptr(), t.numel(), L.dtype_code(t.dtype), ctypes.cast(self.stage_arr, ctypes.c_void_p),
            ctypes.cast(self.flag_arr, ctypes.c_void_p), self.rank, self.world, self.stage_elems, self.blocks,
            self.epoch_done.data_ptr(), L.stream_ptr()), "tp_allreduce_oneshot")

    def fits_rows(self, rows: int, n: int, dtype) -> bool:
        """Rank-invariant admission of the fused projection -> all-reduce -> norm launch."""
        return dtype in (torch.float16, torch.bfloat16) and n % 8 == 0 and n <= 8192 and 0 < rows * n <= self.stage_elems

    def all_reduce_norm_partials(self, parts: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                 out: torch.Tensor) -> None:
        import ctypes

        L = self.L
        s, m, n = parts.shape
        L.check(self.lib.ll_tp_allreduce_norm_partials(
            out.data_ptr(), parts.data_ptr(), s, residual.data_ptr(), weight.data_ptr(), m, n, float(eps),
            L.dtype_code(out.dtype), ctypes.cast(self.stage_arr, ctypes.c_void_p), ctypes.cast(self.flag_arr, ctypes.c_void_p),
            self.rank, self.world, self.stage_elems, self.blocks, self.epoch_done.data_ptr(), L.stream_ptr()),
            "tp_allreduce_norm_partials")

    def error(self) -> int:
        """The device error word (synchronises): bit 0 = a peer's flag did not arrive within the spin bound."""
        import ctypes

        word = ctypes.c_int32()
        self.L.check(self.lib.ll_tp_error_word(self._mine[1], self.flag_words, ctypes.byref(word)), "tp_error_word")
        return int(word.value)

    def close(self) -> None:
        for p in self._opened:
            self.lib.ll_tp_ipc_close(p)
        for p in self._mine:
            self.lib.ll_tp_shared_free(p)
        self._opened, self._mine = [], []


def enable_oneshot_all_reduce(max_elems: int, blocks: int = 64) -> None:
    """Route ``all_reduce_tp`` of 16-bit tensors of up to ``max_elems`` elements through the one-shot peer-to-peer kernel
    (opt-in; every rank of the TP group must call this, with the same arguments, after ``init_parallel``).  Larger /
    other tensors keep the backend's collective."""
    global _ONESHOT
    if _TP_WORLD_SIZE <= 1 or _TP_GROUP is None:
        raise RuntimeError("enable_oneshot_all_reduce needs an initialised tensor-parallel group of more than one rank")
    if _TP_WORLD_SIZE > 8:
        raise ValueError("the one-shot all-reduce serves up to 8 ranks (one xGMI hop)")
    if _ONESHOT is not None:
        _ONESHOT.close()
    _ONESHOT = _OneShot(max_elems, blocks)


def fused_reduce_norm_available(rows: int, n: int, dtype) -> bool:
    """True when a row-parallel projection may hand its split-K partials to the fused all-reduce + add-and-normalise
    launch (one-shot kernel enabled for this TP group, payload fits its staging buffer)."""
    return _TP_WORLD_SIZE > 1 and _ONESHOT is not None and _ONESHOT.fits_rows(rows, n, dtype)


def all_reduce_norm_partials(parts, residual, weight, eps, out) -> None:
    _ONESHOT.all_reduce_norm_partials(parts, residual, weight, eps, out)


def oneshot_error() -> int:
    return 0 if _ONESHOT is None else _ONESHOT.error()


def check_collective_errors() -> None:
    """Raise if the one-shot all-reduce ever lost a peer (sticky device error word; synchronises).  Called by the engine
    at the end of a decode and by the bench: a timed-out flag poisons the activations with NaN on the device, this turns
    it into an exception on the host."""
    if _ONESHOT is not None and _ONESHOT.error():
        raise RuntimeError(f"tensor-parallel rank {_TP_RANK}: a peer's flag of the one-shot all-reduce did not arrive within "
                           "the spin bound; the step's activations were poisoned with NaN (a rank died or stalled)")


destroy_tensor_parallel = destroy_parallel


def get_tp_rank() -> int:
    return _TP_RANK


def get_tp_world_size() -> int:
    return _TP_WORLD_SIZE


def get_dp_rank() -> int:
    return _DP_RANK


def get_dp_world_size() -> int:
    return _DP_WORLD_SIZE


def get_world_size() -> int:
    return _DP_WORLD_SIZE * _TP_WORLD_SIZE


def divide(a: int, b: int, what: str = "") -> int:
    if a % b != 0:
        raise ValueError(f"{what or 'value'} {a} does not divide across {b} tensor-parallel ranks")
    return a // b


def simulate_shard(tp_size: int, rank: int = 0) -> None:
    """Measurement aid (``bench.py --shard-sim``; round-5 review, item 5): build and run ONE rank's shard of a ``tp_size``-way
    tensor-parallel model on one device with no peers and no process group.  Shapes, launch structure and bytes are the
    rank's; every collective is replaced by a same-size local device copy (what the collective moves through this rank), so the
    result is the rank's partial sums -- a timing of one rank's compute, a CEILING for the scaling curve, never a scaling point.
    ``destroy_parallel()`` ends it."""
    global _TP_RANK, _TP_WORLD_SIZE, _SIMULATED
    if dist.is_available() and dist.is_initialized():
        raise RuntimeError("simulate_shard: a process group is live (this mode has no peers)")
    _TP_WORLD_SIZE, _TP_RANK, _SIMULATED = int(tp_size), int(rank), int(tp_size) > 1


def shard_simulated() -> bool:
    return _SIMULATED


def simulated_collective(numel: int, dtype: torch.dtype, device) -> None:
    """The stand-in for one all-reduce of ``numel`` elements in simulate_shard mode: one device-to-device copy of that size
    between two persistent scratch buffers (capturable: the buffers are allocated on first use, outside a capture)."""
    key = (int(numel), dtype, str(device))
    pair = _SIM_SCRATCH.get(key)
    if pair is None:
        pair = _SIM_SCRATCH[key] = (torch.zeros(numel, dtype=dtype, device=device), torch.empty(numel, dtype=dtype, device=device))
    pair[1].copy_(pair[0])


def all_reduce_tp(tensor: torch.Tensor) -> torch.Tensor:
    """In-place SUM over the TP group; identity when ``world_size == 1``."""
    if _SIMULATED:
        simulated_collective(tensor.numel(), tensor.dtype, tensor.device)
        return tensor
    if _TP_WORLD_SIZE <= 1 and not _FORCE_COLLECTIVE:
        return tensor
    if _ONESHOT is not None and _ONESHOT.fits(tensor):
        _ONESHOT.all_reduce(tensor)
        return tensor
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=_TP_GROUP)
    return tensor


def all_reduce_min(value: int) -> int:
    """Smallest ``value`` across the TP group (agreeing on a KV pool size, model_runner.py:84-89)."""
    if _TP_WORLD_SIZE <= 1 or _SIMULATED:
        return value
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_TP_GROUP)
    return int(t.item())


# ------------------------------------------------------------------------------------- #
# multi-GPU start-up checks (round 5; the reference picks NCCL unconditionally, parallel_state.py:44-71, 208-213)
# ------------------------------------------------------------------------------------- #
def collective_decision(local_ok: int, local_reason: str, group=None, gather=None):
    """Group-wide choice between the one-shot peer-to-peer all-reduce and the backend collective (RCCL), from every rank's
    local start-up check: the kernel is used only if EVERY rank passed; otherwise every rank falls back, and every rank
    learns WHY (the first failing rank's reason) so the JSON line of rank 0 can say it.  ``gather``: injection point for tests
    (a callable that returns the list of every rank's ``(ok, reason)``); default ``all_gather_object`` over ``group``.
    Returns ``(use_oneshot, label, reason)`` -- identical on all ranks."""
    mine = (int(bool(local_ok)), str(local_reason or ""))
    if gather is not None:
        everyone = gather(mine)
    elif dist.is_available() and dist.is_initialized():
        everyone = [None] * dist.get_world_size(group)
        dist.all_gather_object(everyone, mine, group=group)
    else:
        everyone = [mine]
    failed = [(r, why) for r, (ok, why) in enumerate(everyone) if not ok]
    if not failed:
        return True, "oneshot (peer-mapped buffers, checked against the backend collective at start-up)", ""
    r, why = failed[0]
    reason = f"rank {r}: {why or 'start-up check failed'}" + (f" (+{len(failed) - 1} more rank(s))" if len(failed) > 1 else "")
    return False, "rccl (one-shot kernel failed its start-up check)", reason


def peer_access_matrix():
    """``hipDeviceCanAccessPeer`` for every ordered pair of visible devices (row = accessing device): what the one-shot
    all-reduce needs over xGMI.  Printed by tools/multi_gpu.sh before any bench."""
    n = torch.cuda.device_count()
    return [[1 if i == j else int(torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]


def assert_distinct_devices(local_device: int, world: int, group=None, gather=None):
    """On a box that HAS ``world`` devices every rank must sit on its own (a launcher that pins all ranks to device 0 would
    silently measure a time-sliced GPU).  Returns the list of (host, device, uuid) per rank; raises on a collision unless the
    debugging set-up (``LL_BENCH_DEVICE`` / fewer devices than ranks) was asked for."""
    import socket
    ident = (socket.gethostname(), int(local_device), _device_uuid(local_device))
    if gather is not None:
        everyone = gather(ident)
    elif dist.is_available() and dist.is_initialized():
        everyone = [None] * dist.get_world_size(group)
        dist.all_gather_object(everyone, ident, group=group)
    else:
        everyone = [ident]
    shared_ok = bool(os.environ.get("LL_BENCH_DEVICE")) or (torch.cuda.is_available() and torch.cuda.device_count() < world)
    keys = [(h, u or d) for h, d, u in everyone]
    if len(set(keys)) != len(keys) and not shared_ok:
        raise RuntimeError(f"{world} ranks but only {len(set(keys))} distinct devices: {everyone} -- one rank per GPU expected "
                           "(LOCAL_RANK -> device); set LL_BENCH_DEVICE to share a device on purpose")
    return everyone


def _device_uuid(index: int) -> str:
    try:
        return str(torch.cuda.get_device_properties(index).uuid)
    except Exception:
        return ""
