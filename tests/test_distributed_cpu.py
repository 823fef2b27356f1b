"""TP path on CPU: world_size-2 ``gloo`` process groups (the GPU box runs the same code on RCCL).

Covers the rank grid arithmetic (reference tests/distributed/test_parallel_state.py), the in-place
SUM all-reduce of RowParallelLinear, shard-alignment errors, and that column->row sharded
projections reproduce the unsharded result (which the reference never unit-tests for tp > 1)."""

import math
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


def test_moe_expert_scale_grid_is_refined_before_an_inside_block_cut(monkeypatch):
    """Extension (SURVEY 8e, config 5 at TP 4 / 8): Qwen3-30B-A3B's 768 expert channels over 4 / 8 ranks = 192 / 96 per
    rank cut the checkpoint's 128 x 128 fp8 scale blocks.  The loader refines the grid to gcd(128, shard) = 64 / 32 channels
    (every scale repeated) and cuts evenly; the dequantised shard is exactly the rank's slice of the dequantised tensor --
    for gate / up (cut along N, dim 0 of the grid) and for down (cut along K, dim 1)."""
    import lite_llama_amd.weights as W

    g = torch.Generator().manual_seed(3)
    I, H = 768, 512
    for world in (4, 8):
        sh = I // world
        cut = math.gcd(128, sh)
        for which, (n, k, dim) in {"gate": (I, H, 0), "down": (H, I, 1)}.items():
            w = torch.randn(n, k, generator=g)
            s = torch.rand(n // 128, k // 128, generator=g) + 0.5
            full = w * s.repeat_interleave(128, 0).repeat_interleave(128, 1)
            name = "layers.0.mlp.experts.gate_up_proj_scale_inv" if which == "gate" else "layers.0.mlp.experts.down_proj_scale_inv"
            for rank in range(world):
                monkeypatch.setattr(W, "get_tp_world_size", lambda world=world: world)
                monkeypatch.setattr(W, "get_tp_rank", lambda rank=rank: rank)
                s_r = W._narrow_for_rank(name, s, dim)
                w_r = w.narrow(dim, rank * sh, sh)
                gn, gk = (cut, 128) if dim == 0 else (128, cut)
                assert tuple(s_r.shape) == (w_r.shape[0] // gn, w_r.shape[1] // gk)
                deq = w_r * s_r.repeat_interleave(gn, 0).repeat_interleave(gk, 1)
                assert torch.equal(deq, full.narrow(dim, rank * sh, sh)), (world, which, rank)
    # aligned cuts are untouched (TP = 2: three whole blocks per rank)
    monkeypatch.setattr(W, "get_tp_world_size", lambda: 2)
    monkeypatch.setattr(W, "get_tp_rank", lambda: 1)
    s = torch.rand(6, 4, generator=g)
    assert torch.equal(W._narrow_for_rank("layers.0.mlp.experts.gate_up_proj_scale_inv", s, 0), s[3:6])


def test_moe_block_shapes_under_an_inside_block_cut(monkeypatch):
    """SparseMoeBlock at TP = 4 / 8 with 128 x 128 fp8 blocks: scale_cut 64 / 32, gate|up's scale grid refined along N, down's
    along K; int4 experts and one-sided groups still refuse the cut with the reference's message."""
    import lite_llama_amd.model as M
    from lite_llama_amd.quantization import QuantConfig

    geo = M.tiny_geometry(hidden_size=512, intermediate_size=1024, num_layers=1, num_heads=8, num_kv_heads=4, head_dim=64,
                          vocab_size=256, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=768)
    for world, cut in ((2, 0), (4, 64), (8, 32)):
        monkeypatch.setattr(M, "get_tp_world_size", lambda world=world: world)
        blk = M.SparseMoeBlock(geo, QuantConfig.fp8_block(128, 128))
        sh = 768 // world
        assert blk.scale_cut == cut and blk.moe_intermediate_size == sh
        g = cut or 128
        assert tuple(blk.experts["gate_up_proj_scale_inv"].shape) == (4, 2 * sh // g, 512 // 128)
        assert tuple(blk.experts["down_proj_scale_inv"].shape) == (4, 512 // 128, sh // g)
    monkeypatch.setattr(M, "get_tp_world_size", lambda: 4)
    with pytest.raises(ValueError, match="not a multiple of the"):
        M.SparseMoeBlock(geo, QuantConfig.int8_groupwise(128))
    # MoE geometries now get an attention-head plan: Qwen3-30B-A3B's 32 / 4 heads at TP = 8 replicate every KV head on two ranks
    plan = M.shard_plan(M.GEOMETRY["qwen3-30b-a3b"], QuantConfig.fp8_per_channel(), tp=8)
    assert plan is not None and [n for _, n in plan.q_heads] == [4] * 8 and [s for s, _ in plan.kv_heads] == [0, 0, 1, 1, 2, 2, 3, 3]
    assert M.shard_plan(M.GEOMETRY["qwen3-30b-a3b"], QuantConfig.fp8_per_channel(), tp=4) is None  # the reference's equal cuts


# ------------------------------------------------------------------------------------------------------------------ #
# Multi-GPU start-up checks (round 5): the one-shot / RCCL decision and the one-rank-per-device assertion
# ------------------------------------------------------------------------------------------------------------------ #
def _decision_worker(rank, world, port, q):
    import lite_llama_amd.distributed.parallel_state as ps
    os.environ["LL_DIST_BACKEND"] = "gloo"
    try:
        ps.init_tensor_parallel(rank, world, master_port=port)
        # case 1: every rank passed -> the kernel; case 2: rank 1 failed -> BOTH ranks fall back and both know why
        a = ps.collective_decision(1, "", group=ps._TP_GROUP)
        b = ps.collective_decision(0 if rank == 1 else 1, "sums differ from RCCL's" if rank == 1 else "", group=ps._TP_GROUP)
        # two ranks on one host with the same (absent) device identity: a collision unless sharing was asked for
        try:
            ps.assert_distinct_devices(0, world, group=ps._TP_GROUP)
            collided = False
        except RuntimeError:
            collided = True
        q.put((rank, a, b, collided))
    except Exception as exc:  # pragma: no cover
        q.put((rank, repr(exc), None, None))
    finally:
        ps.destroy_parallel()


def test_collective_fallback_decision_is_group_wide_and_carries_the_reason():
    import lite_llama_amd.distributed.parallel_state as ps

    # pure form (injected gather): all ok / one failure / several failures
    use, label, why = ps.collective_decision(1, "", gather=lambda m: [m, (1, ""), (1, "")])
    assert use and label.startswith("oneshot") and why == ""
    use, label, why = ps.collective_decision(1, "", gather=lambda m: [m, (0, "a peer flag timed out"), (1, "")])
    assert not use and label.startswith("rccl") and why == "rank 1: a peer flag timed out"
    use, label, why = ps.collective_decision(0, "RuntimeError: hipIpcOpenMemHandle", gather=lambda m: [m, (0, "x"), (1, "")])
    assert not use and why.startswith("rank 0: RuntimeError: hipIpcOpenMemHandle") and "+1 more" in why
    # one rank per device: distinct identities pass, a collision raises, the debugging set-up is allowed through
    ok = ps.assert_distinct_devices(0, 2, gather=lambda m: [("h", 0, "uuid-a"), ("h", 1, "uuid-b")])
    assert len(ok) == 2
    if not torch.cuda.is_available():  # (on a GPU box with fewer devices than ranks sharing is the documented fallback)
        with pytest.raises(RuntimeError, match="distinct devices"):
            ps.assert_distinct_devices(0, 2, gather=lambda m: [("h", 0, "uuid-a"), ("h", 0, "uuid-a")])
    os.environ["LL_BENCH_DEVICE"] = "0"
    try:
        ps.assert_distinct_devices(0, 2, gather=lambda m: [("h", 0, "uuid-a"), ("h", 0, "uuid-a")])
    finally:
        os.environ.pop("LL_BENCH_DEVICE")
    # over a real gloo group of two
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decision_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, a, b, collided in results:
        assert a[0] is True and a[2] == "", (rank, a)
        assert b[0] is False and b[1].startswith("rccl") and b[2] == "rank 1: sums differ from RCCL's", (rank, b)
        assert collided is (not torch.cuda.is_available())
