"""TP path on CPU: world_size-2 ``gloo`` process groups (the GPU box runs the same code on RCCL).

Covers the rank grid arithmetic (reference tests/distributed/test_parallel_state.py), the in-place
SUM all-reduce of RowParallelLinear, shard-alignment errors, and that column->row sharded
projections reproduce the unsharded result (which the reference never unit-tests for tp > 1)."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from lite_llama_amd.distributed import parallel_state as ps
from lite_llama_amd.quantization import QuantConfig


def test_grid_coordinates():
    assert ps.grid_coordinates(0, 2, 2) == (0, 0)
    assert ps.grid_coordinates(3, 2, 2) == (1, 1)
    assert ps.grid_coordinates(5, 2, 3) == (2, 1)
    with pytest.raises(ValueError):
        ps.grid_coordinates(4, 2, 2)
    with pytest.raises(ValueError):
        ps.grid_coordinates(0, 0, 1)


def test_world_of_one_is_identity():
    ps.init_parallel(0, 1, 1)
    t = torch.ones(3)
    assert ps.all_reduce_tp(t) is t
    assert ps.all_reduce_min(7) == 7
    assert ps.get_tp_world_size() == 1 and ps.get_tp_rank() == 0
    assert ps.divide(8, 2) == 4
    with pytest.raises(ValueError):
        ps.divide(7, 2, "heads")
    ps.destroy_parallel()


def test_shard_alignment_rules():
    q4 = QuantConfig.int4_groupwise(128)
    assert q4.shard_is_aligned(4736) and not q4.shard_is_aligned(2368)  # Qwen2.5-7B: TP4 ok, TP8 cuts a group
    fp8 = QuantConfig.fp8_block()
    assert fp8.shard_is_aligned(384) and not fp8.shard_is_aligned(192)  # Qwen3-30B-A3B: TP2 ok, TP4 not
    assert QuantConfig.int8_per_channel().shard_is_aligned(37)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        ps.init_tensor_parallel(rank, world, master_port=port)
        from lite_llama_amd.linear import ColumnParallelLinear, RowParallelLinear

        assert ps.get_tp_world_size() == world and ps.get_tp_rank() == rank
        # all_reduce_tp: in place SUM
        t = torch.full((4,), float(rank + 1))
        r = ps.all_reduce_tp(t)
        assert r is t and torch.all(t == 3.0)
        assert ps.all_reduce_min(10 + rank) == 10
        # column -> row sharded pair == full product
        torch.manual_seed(0)
        hid, inter = 64, 256
        w1 = (torch.randn(inter, hid) * 0.1).half()
        w2 = (torch.randn(hid, inter) * 0.1).half()
        x = (torch.randn(5, hid)).half()
        col = ColumnParallelLinear(hid, inter)
        row = RowParallelLinear(inter, hid)
        col.weight.data.copy_(w1.chunk(world, dim=0)[rank])
        row.weight.data.copy_(w2.chunk(world, dim=1)[rank])
        y = row(col(x).float().half())
        full = ((x.float() @ w1.float().T).half().float() @ w2.float().T)
        ok = torch.allclose(y.float(), full, rtol=3e-2, atol=3e-2)
        # misaligned shard is rejected before any weight is created
        try:
            RowParallelLinear(128 * 3, 64, quant=QuantConfig.int4_groupwise(128))
            aligned_err = False
        except ValueError:
            aligned_err = True
        q.put((rank, ok, aligned_err))
    except Exception as exc:  # pragma: no cover
        q.put((rank, False, repr(exc)))
    finally:
        ps.destroy_parallel()


def test_tp2_gloo_allreduce_and_sharded_linears():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, extra in results:
        assert ok is True, (rank, extra)
        assert extra is True  # 192-wide shard of a 384-wide int4 layer cuts a 128-group -> ValueError
