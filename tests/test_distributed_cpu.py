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


# ------------------------------------------------------------------------------------------------------------------ #
# Extension plans (distributed/partition.py): TP beyond the reference's divisibility rules -- SURVEY 8e
# ------------------------------------------------------------------------------------------------------------------ #
def test_shard_plan_arithmetic():
    from lite_llama_amd.distributed.partition import admissible_tp, make_plan

    # the reference's own cases stay the reference's equal cuts
    for hq, hkv, inter, tp in [(28, 4, 18944, 1), (28, 4, 18944, 2), (28, 4, 18944, 4), (32, 8, 14336, 8), (12, 2, 8960, 2)]:
        p = make_plan(hq, hkv, 128, inter, tp)
        assert p.uniform
        assert p.q_heads == tuple((r * (hq // tp), hq // tp) for r in range(tp))
        assert p.inter == tuple((r * (inter // tp), inter // tp) for r in range(tp))
    # Qwen2.5-7B at TP = 8: every KV head on two ranks that split its seven query heads 4 + 3; 148 int4 groups = 4 x 19 + 4 x 18
    p = make_plan(28, 4, 128, 18944, 8)
    assert not p.uniform
    covered = [h for s, n in p.q_heads for h in range(s, s + n)]
    assert covered == list(range(28))                                   # every query head exactly once, in order
    for r, ((qs, qn), (ks, kn)) in enumerate(zip(p.q_heads, p.kv_heads)):
        assert kn == 1 and all(h // 7 == ks for h in range(qs, qs + qn))  # a rank's query heads belong to ITS KV head
    assert sorted(n for _, n in p.q_heads) == [3] * 4 + [4] * 4
    assert sum(n for _, n in p.inter) == 18944 and all(s % 128 == 0 and n % 128 == 0 for s, n in p.inter)
    assert [s for s, _ in p.inter] == [sum(n for _, n in p.inter[:r]) for r in range(8)]  # contiguous
    assert max(n for _, n in p.inter) - min(n for _, n in p.inter) == 128
    assert admissible_tp(28, 4, 128, 18944, 8) == 8 and admissible_tp(28, 4, 128, 18944, 6) == 2
    with pytest.raises(ValueError):
        make_plan(28, 4, 128, 18944, 3)          # neither tp | Hkv nor Hkv | tp
    with pytest.raises(ValueError):
        make_plan(8, 4, 128, 1024, 16)           # two query heads per KV head cannot be shared by four ranks
    with pytest.raises(ValueError):
        make_plan(28, 4, 128, 18944 + 64, 8)     # intermediate not made of whole groups


def _plan_worker(rank, world, port, q):
    """The decoder block's two sharded halves in plain fp32 torch arithmetic on the PRODUCT's shard layout (model.py
    builds the rank's modules from the plan, init_synthetic cuts the seeded full matrices): attention with this rank's
    query heads over its (replicated) KV head -> o_proj partial -> all-reduce; gate/up -> silu*up -> down partial ->
    all-reduce.  The parent compares with the same arithmetic on the unsharded matrices."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        ps.init_tensor_parallel(rank, world, master_port=port)
        out = _block_math(*_plan_case())
        q.put((rank, True, out.numpy()))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, False, repr(exc) + traceback.format_exc()[-1200:]))
    finally:
        ps.destroy_parallel()


def _plan_case():
    from lite_llama_amd.model import CausalLM, tiny_geometry

    geo = tiny_geometry(hidden_size=64, intermediate_size=128 * 12, num_layers=1, num_heads=28, num_kv_heads=4, head_dim=16,
                        vocab_size=32, qkv_bias=True)
    model = CausalLM(geo).init_synthetic(seed=3, quant=None, device="cpu")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, geo.hidden_size, generator=g)
    return geo, model, x


def _block_math(geo, model, x):
    layer = model.layers[0]
    at, mlp = layer.self_attn, layer.mlp
    d = geo.head_dim
    q = (x @ at.q_proj.weight.float().T + at.q_proj.bias.float()).view(-1, at.num_heads, d)
    kv = (x @ at.kv_proj.weight.float().T + at.kv_proj.bias.float()).view(-1, 2 * at.num_kv_heads, d)
    k, v = kv[:, : at.num_kv_heads], kv[:, at.num_kv_heads:]
    group = at.num_heads // at.num_kv_heads
    heads = []
    for h in range(at.num_heads):  # every token attends to all 5 tokens (no mask: the cut, not causality, is under test)
        kk, vv = k[:, h // group], v[:, h // group]
        heads.append(torch.softmax(q[:, h] @ kk.T / d ** 0.5, -1) @ vv)
    attn = torch.stack(heads, 1).reshape(-1, at.num_heads * d)
    o = ps.all_reduce_tp(attn @ at.o_proj.weight.float().T)
    h1 = x + o
    act = torch.nn.functional.silu(h1 @ mlp.gate_proj.weight.float().T) * (h1 @ mlp.up_proj.weight.float().T)
    return h1 + ps.all_reduce_tp(act @ mlp.down_proj.weight.float().T)


def test_tp8_extension_plan_sharded_equals_unsharded_gloo():
    """Eight gloo ranks on the extension plan (28 / 4 heads -> 4 + 3 query heads per rank, each KV head on two ranks; twelve
    128-channel groups -> 2,2,2,2,1,1,1,1) reproduce the unsharded block; every rank holds ONE KV head."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, ok, out in results:
        assert ok is True, (rank, out)
    want = _block_math(*_plan_case())  # tp = 1 in this process: the same seeded full matrices, uncut
    for rank, _, out in results:
        torch.testing.assert_close(torch.from_numpy(out), want, rtol=1e-4, atol=1e-4)
