"""Device placement of the multi-process GPU tests: one device per rank over RCCL (backend "nccl") when the box has at
least ``world`` devices -- the set-up of a real tensor-parallel run, peer mappings crossing xGMI -- else every rank on device
0 with the gloo host channel for the collectives / handle exchange (RCCL refuses two ranks on one device): the sharded
kernels and the one-shot kernel's IPC mappings still run for real, time-slicing the one GPU."""
import os

import torch


def place_rank(rank: int, world: int):
    """-> (device index, True when every rank has its own device).  Sets / clears LL_DIST_BACKEND accordingly."""
    if os.environ.get("LL_TEST_FORCE_SHARED_DEVICE") or torch.cuda.device_count() < world:
        os.environ["LL_DIST_BACKEND"] = "gloo"
        torch.cuda.set_device(0)
        return 0, False
    os.environ.pop("LL_DIST_BACKEND", None)
    torch.cuda.set_device(rank)
    return rank, True


def assert_real_multi_gpu(ps, distinct: bool) -> None:
    """On a multi-device box the collectives must be RCCL's and the one-shot kernel's peer buffers on distinct devices."""
    if not distinct:
        return
    assert ps._backend() == "nccl", ps._backend()
    if ps._ONESHOT is not None:
        assert ps._ONESHOT.crosses_devices(), ps._ONESHOT.peer_devices
