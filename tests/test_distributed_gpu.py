"""Tensor-parallel decode on the device: two ranks share the one GPU of the box (gloo carries the
all-reduces through the host -- RCCL refuses two ranks on one device), so the SHARDED int4 kernels,
the fused q|k|v / gate|up launches on shard shapes and the per-rank KV pools run for real.  The
tp = 2 model is an exact partition of the tp = 1 model (same seed, row shards keep their groups, column
shards cut at group boundaries), so logits agree up to the fp16 all-reduce's summation order.
"""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _geometry(which="tp2"):
    from lite_llama_amd.model import tiny_geometry

    if which == "tp4plan":  # 14 / 2 heads at TP = 4: each KV head on two ranks (4 + 3 query heads), six 128-groups -> 2, 2, 1, 1
        return tiny_geometry(hidden_size=512, intermediate_size=768, num_layers=2, num_heads=14, num_kv_heads=2, head_dim=128,
                             vocab_size=640, qkv_bias=True)
    if which == "tp8":  # Qwen2.5-7B's head layout (28 / 4 heads of 128) and an intermediate of twelve 128-groups: outside the
        # reference's rules at TP = 8 -> extension plan (4 + 3 query heads per rank, KV heads on two ranks, 2,2,2,2,1,1,1,1 groups)
        return tiny_geometry(hidden_size=512, intermediate_size=1536, num_layers=2, num_heads=28, num_kv_heads=4, head_dim=128,
                             vocab_size=640, qkv_bias=True)
    if which == "moe8":  # Qwen3-MoE-shaped: 16 / 4 heads with q/k norms (KV heads on two ranks each at TP = 8), 8 experts top-2
        # whose 256 intermediate channels are cut into 32 per rank
        return tiny_geometry(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=16, num_kv_heads=4, head_dim=64,
                             vocab_size=640, qkv_bias=False, use_qk_norm=True, num_experts=8, num_experts_per_tok=2,
                             moe_intermediate_size=256, norm_topk_prob=True)
    return tiny_geometry(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2, head_dim=64,
                         vocab_size=640, qkv_bias=True)


def _run(which="tp2"):
    """prefill 2 sequences + 6 greedy decode steps (eager); returns (first tokens, tokens, logits of the first decode step)."""
    from lite_llama_amd.executor import DecodeEngine
    from lite_llama_amd.model import CausalLM
    from lite_llama_amd.quantization import QuantConfig

    quant = None if which == "moe8" else QuantConfig.int4_groupwise(128)  # (fp16 experts: every rank's shard is an exact cut)
    model = CausalLM(_geometry(which), quant).init_synthetic(seed=9, quant=quant, device="cuda")
    eng = DecodeEngine(model, max_batch=2, max_seq_len=32)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 640, (2, 7), generator=g).cuda()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        torch.cuda.synchronize()
        dist.barrier()  # ranks sharing one GPU build their models seconds apart; the one-shot kernel's patience is ~10 s
    first = eng.prefill(ids, torch.tensor([7, 5], device="cuda"))
    grabbed = []
    orig = model.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        grabbed.append(out.detach().float().cpu())
        return out

    model.forward = spy
    toks = eng.decode(first, 6, use_graph=False)
    model.forward = orig
    return first.cpu(), toks.cpu(), grabbed[0]


def _worker(rank, world, port, q, which="tp2", oneshot=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LL_TP_SPIN_LOG2"] = "28"  # ranks time-slicing ONE GPU: the late rank gets 1 / N of the device while the others spin
    from lite_llama_amd.distributed import parallel_state as ps
    from tests._dist import assert_real_multi_gpu, place_rank

    try:
        _, distinct = place_rank(rank, world)  # one device per rank over RCCL when the box has them, else device 0 + gloo
        ps.init_tensor_parallel(rank, world, master_port=port)
        assert ps.get_tp_world_size() == world
        if oneshot:  # the collective of the real multi-GPU run: one-shot kernel over IPC mappings, fused with the norms
            import torch.distributed as dist
            ps.enable_oneshot_all_reduce(2 * 7 * 512)
            dist.barrier()
        assert_real_multi_gpu(ps, distinct)
        first, toks, logits = _run(which)
        if oneshot:
            assert ps.oneshot_error() == 0
        q.put((rank, True, first.numpy(), (toks.numpy(), logits.numpy())))  # by value: a shared-memory tensor handle dies with the worker
    except Exception as exc:  # pragma: no cover
        import traceback

        q.put((rank, False, repr(exc) + traceback.format_exc()[-1500:], None))
    finally:
        ps.destroy_parallel()


@pytest.mark.gpu
def test_tp2_sharded_int4_decode_matches_tp1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, ok, a, b in results:
        assert ok is True, (rank, a)
    t = torch.from_numpy
    first0, (toks0, logits0) = t(results[0][2]), tuple(map(t, results[0][3]))
    first1, (toks1, logits1) = t(results[1][2]), tuple(map(t, results[1][3]))
    assert torch.equal(first0, first1) and torch.equal(toks0, toks1)  # every rank computes the same argmax
    assert torch.equal(logits0, logits1)
    ref_first, ref_toks, ref_logits = _run()                           # tp = 1 in this process
    assert torch.equal(first0, ref_first)
    # LOGITS of the first decode step (same context on both sides) at a stated tolerance: the tp = 2 model is an
    # exact partition of the tp = 1 model, the only difference is the fp16 rounding of the two partial sums
    # before the all-reduce (2 per layer) -- 1e-2 absolute on logits of magnitude ~1
    torch.testing.assert_close(logits0, ref_logits, rtol=1e-2, atol=1e-2)
    # greedy paths agree while the top-2 margin is not inside that noise
    agree = (toks0 == ref_toks).float().mean().item()
    assert agree >= 0.75, (toks0, ref_toks)


@pytest.mark.gpu
@pytest.mark.parametrize("world,which,oneshot", [(8, "tp8", False), (4, "tp4plan", True), (8, "moe8", False)])
def test_tp8_extension_plan_int4_decode_matches_tp1(world, which, oneshot):
    """EIGHT ranks on the one GPU, a geometry the reference refuses at TP = 8 (28 / 4 heads, an intermediate that does not
    divide into eight group-aligned parts): the extension plan of distributed/partition.py -- 4 + 3 query heads per rank,
    every KV head replicated on two ranks (each rank's pool holds ONE KV head), whole 128-groups per rank -- decodes the
    tokens of the unsharded model; the first decode step's logits agree at 1e-2 (sixteen fp16 partial sums per layer).
    ``oneshot``: the all-reduces on the one-shot peer-to-peer kernel, the decode step's row-parallel projections through the
    fused partials + all-reduce + add-and-normalise launch (what ``bench.py --gpus 8`` runs), on uneven shards -- with four
    ranks (14 / 2 heads, six groups): eight spinning ranks time-slicing one GPU starve each other past any patience."""
    from lite_llama_amd.distributed.partition import make_plan

    assert not make_plan(28, 4, 128, 1536, 8).uniform and not make_plan(14, 2, 128, 768, 4).uniform
    # "moe8": a sparse-MoE geometry at TP = 8 -- the attention heads on the plan (the reference refuses tp > Hkv), the
    # experts' intermediate dimension on the equal cut
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, which, oneshot)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, ok, a, b in results:
        assert ok is True, (rank, a)
    t = torch.from_numpy
    ref_first, ref_toks, ref_logits = _run(which)  # tp = 1 in this process
    for rank, _, first, (toks, logits) in results:
        assert torch.equal(t(first), t(results[0][2])) and torch.equal(t(toks), t(results[0][3][0]))
        assert torch.equal(t(logits), t(results[0][3][1]))
    assert torch.equal(t(results[0][2]), ref_first)
    torch.testing.assert_close(t(results[0][3][1]), ref_logits, rtol=1e-2, atol=1e-2)
    agree = (t(results[0][3][0]) == ref_toks).float().mean().item()
    assert agree >= 0.75, (results[0][3][0], ref_toks)
