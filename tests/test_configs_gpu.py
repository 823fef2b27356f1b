"""GPU tests at the WORKLOAD shapes of BASELINE.json's secondary configurations, through the C ABI:

* config 5 -- Qwen3-30B-A3B-FP8: fused_moe with fp8-e4m3 experts and 128 x 128 scale blocks at hidden 2048 /
  moe_intermediate 768 / 128 experts / top-8, one routed block of the model at full width, and its TP = 2 split
  (two ranks on the one GPU, gloo all-reduce) -- reference kernels/fused_moe.py:352-438, models/qwen3_moe.py:60-111;
* config 2 -- Qwen2.5-1.5B bf16 (12 / 2 heads of 128, batch 32, context 576): the bf16 attention / norm / rope
  kernels inside a captured graph (the linears of the reference are fp16-only, executor/loader.py:32);
* config 4 -- Llama-3-8B SmoothQuant W8A8 at batch 32: every projection width of the model incl. the 14336-wide MLP;
* RCCL: a decode step whose row-parallel projections really call the backend all-reduce, captured in a hipGraph
  (one rank, world size 1 -- what a one-GPU box can prove ahead of the 8-GPU run).
"""

import math
import os
import socket
import types

import pytest
import torch
import torch.multiprocessing as mp

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def K():
    import lite_llama_amd.kernels as k

    return k


def close(a, b, tol):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), rtol=tol, atol=tol)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ------------------------------------------------------------------------------------- #
# config 5: fp8-block experts at the 30B-A3B widths
# ------------------------------------------------------------------------------------- #
H5, I5, E5, TOPK5 = 2048, 768, 128, 8


def _fp8_experts(seed, inter=I5, rows=None):
    """Random e4m3 expert weights + 128 x 128 block scales in the checkpoint format (weight_scale_inv)."""
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(E5, 2 * inter, H5, generator=g) * 0.6).to(torch.float8_e4m3fn).view(torch.uint8)
    w2 = (torch.randn(E5, H5, inter, generator=g) * 0.6).to(torch.float8_e4m3fn).view(torch.uint8)
    s1 = torch.rand(E5, 2 * inter // 128, H5 // 128, generator=g) * 0.02 + 0.01
    s2 = torch.rand(E5, H5 // 128, inter // 128, generator=g) * 0.02 + 0.01
    return w1, w2, s1, s2


@pytest.mark.parametrize("T", [1, 64])
def test_config5_fused_moe_fp8_block_vs_oracle(T):
    g = torch.Generator().manual_seed(50 + T)
    w1, w2, s1, s2 = _fp8_experts(5)
    x = (torch.randn(T, H5, generator=g) / H5**0.5 * 4).half()
    ids = torch.stack([torch.randperm(E5, generator=g)[:TOPK5] for _ in range(T)])
    wts = torch.softmax(torch.randn(T, TOPK5, generator=g), dim=-1).half()
    out = K().fused_moe(x.to(DEV), w1.to(DEV), w2.to(DEV), wts.to(DEV), ids.to(DEV), w1_scale=s1.to(DEV),
                        w2_scale=s2.to(DEV), group_n=128, group_k=128)
    assert out.shape == (T, H5) and torch.isfinite(out).all()
    sample = list(range(T)) if T <= 4 else [0, 17, 31, 63]  # the oracle dequantises one expert per slot
    ref = O.fused_moe(x[sample], w1, w2, wts[sample], ids[sample], w1_scale=s1, w2_scale=s2, group_n=128, group_k=128)
    close(out[sample], ref, 2e-2)


def _moe_block(tp_rank=0, tp=1, seed=7):
    """SparseMoeBlock of the 30B-A3B geometry with fp8-block experts; rank ``tp_rank`` of ``tp`` keeps the reference's
    shard (qwen3_moe.py:60-84: gate and up rows and the down columns of its intermediate slice)."""
    from lite_llama_amd.model import SparseMoeBlock, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from lite_llama_amd.quantization.methods import RawParameter

    geo = tiny_geometry(hidden_size=H5, intermediate_size=6144, num_layers=1, num_heads=32, num_kv_heads=4, head_dim=128,
                        vocab_size=512, use_qk_norm=True, num_experts=E5, num_experts_per_tok=TOPK5,
                        moe_intermediate_size=I5, norm_topk_prob=True)
    blk = SparseMoeBlock(geo, QuantConfig.fp8_block(128, 128))
    w1, w2, s1, s2 = _fp8_experts(seed)
    g = torch.Generator().manual_seed(seed + 1)
    gate_w = (torch.randn(E5, H5, generator=g) * 0.05).half()
    sh = I5 // tp
    lo, hi = tp_rank * sh, (tp_rank + 1) * sh
    w1s = torch.cat([w1[:, lo:hi], w1[:, I5 + lo:I5 + hi]], dim=1).contiguous()
    w2s = w2[:, :, lo:hi].contiguous()
    # the checkpoint's 128-blocks on the grid the shard is stored on: 128 when the cut ends on block boundaries (TP 1 / 2),
    # gcd(128, shard) = 64 / 32 at TP 4 / 8 (every scale repeated -- weights.py::expand_scale_grid -- then cut)
    from lite_llama_amd.weights import expand_scale_grid
    cut = blk.scale_cut or 128
    assert cut == math.gcd(128, sh)
    s1f, s2f = expand_scale_grid(s1, 1, 128, cut), expand_scale_grid(s2, 2, 128, cut)
    s1s = torch.cat([s1f[:, lo // cut:hi // cut], s1f[:, (I5 + lo) // cut:(I5 + hi) // cut]], dim=1).contiguous()
    s2s = s2f[:, :, lo // cut:hi // cut].contiguous()
    assert tuple(blk.experts["gate_up_proj_scale_inv"].shape) == tuple(s1s.shape), (blk.experts["gate_up_proj_scale_inv"].shape, s1s.shape)
    assert tuple(blk.experts["down_proj_scale_inv"].shape) == tuple(s2s.shape)
    blk.gate_weight.data = gate_w.to(DEV)
    blk.experts["gate_up_proj"] = RawParameter(w1s.to(DEV))
    blk.experts["gate_up_proj_scale_inv"] = RawParameter(s1s.to(DEV))
    blk.experts["down_proj"] = RawParameter(w2s.to(DEV))
    blk.experts["down_proj_scale_inv"] = RawParameter(s2s.to(DEV))
    return blk, (w1, w2, s1, s2, gate_w)


def _moe_input(T=64):
    g = torch.Generator().manual_seed(99)
    return (torch.randn(T, H5, generator=g) / H5**0.5 * 4).half()


def test_config5_full_width_moe_block_vs_oracle():
    """Router (fp16 linear -> fp32 softmax over all experts -> top-8 -> renormalise) + fp8-block experts of ONE block of
    Qwen3-30B-A3B at batch 64 against the oracle composition on a token sample."""
    blk, (w1, w2, s1, s2, gate_w) = _moe_block()
    x = _moe_input()
    with torch.no_grad():
        out = blk(x.to(DEV))
    sample = [0, 9, 33, 63]
    logits = (x[sample].float() @ gate_w.float().T).half()  # fp16 linear
    probs = torch.softmax(logits, dim=-1, dtype=torch.float32)
    w, ids = torch.topk(probs, TOPK5, dim=-1)
    w = w / w.sum(dim=-1, keepdim=True)
    ref = O.fused_moe(x[sample], w1, w2, w.half(), ids, w1_scale=s1, w2_scale=s2, group_n=128, group_k=128)
    with torch.no_grad():
        w_hip, ids_hip = blk._route(x.to(DEV))[:2]
    assert torch.equal(ids_hip[sample].cpu().sort(-1).values, ids.sort(-1).values)  # same experts chosen
    close(out[sample], ref, 2e-2)


def _tp_moe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LL_DIST_BACKEND"] = "gloo"
    from lite_llama_amd.distributed import parallel_state as ps

    try:
        torch.cuda.set_device(0)
        ps.init_tensor_parallel(rank, world, master_port=port)
        blk, _ = _moe_block(tp_rank=rank, tp=world)
        assert blk.moe_intermediate_size == I5 // world
        with torch.no_grad():
            out = blk(_moe_input().to(DEV))
        q.put((rank, True, out.cpu().numpy()))  # by value: a shared-memory tensor handle dies with the worker
    except Exception as exc:  # pragma: no cover
        import traceback

        q.put((rank, False, repr(exc) + traceback.format_exc()[-1500:]))
    finally:
        ps.destroy_parallel()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config5_moe_block_tp2_matches_tp1(world):
    """The routed block sharded over tensor-parallel ranks against the unsharded block: same values up to the fp16
    rounding of the partial sums.  TP = 2: moe_intermediate 768 -> 384 per rank = 3 scale blocks, one all-reduce per block
    as in qwen3_moe.py:102-111.  TP = 4 / 8 (extension; the reference refuses them): 192 / 96 channels per rank cut the
    128 x 128 scale blocks -- the shard carries the blocks' scales on a 64- / 32-channel grid (gate|up along N, down
    along K: the per-8-k scale path of the grouped GEMM), the ranks still partition the unsharded model exactly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_moe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, ok, payload in results:
        assert ok is True, (rank, payload)
    results = [(r, ok, torch.from_numpy(a)) for r, ok, a in results]
    for r in results[1:]:
        assert torch.equal(results[0][2], r[2])  # every rank holds the reduced sum
    blk, _ = _moe_block()
    with torch.no_grad():
        ref = blk(_moe_input().to(DEV)).cpu()
    scale = ref.float().abs().max().item()
    tol = 4e-3 * max(1.0, world / 2)  # fp16 roundings of `world` partial sums
    assert (results[0][2].float() - ref.float()).abs().max().item() <= tol * scale + tol


# ------------------------------------------------------------------------------------- #
# config 2: bf16 attention / norm / rope at the Qwen2.5-1.5B decode shape, captured
# ------------------------------------------------------------------------------------- #
def test_config2_bf16_decode_kernels_in_a_captured_graph():
    """12 query / 2 KV heads of 128, batch 32, context 576, bf16: rope + KV scatter, flash-decoding and the
    add-and-normalise, captured in one hipGraph; replay == eager (bytes) and both match the oracle."""
    from lite_llama_amd.kernels.norm_act import rope_and_cache
    HQ, HKV, D, B, CTX, HID = 12, 2, 128, 32, 576, 1536
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(2)
    rows = B * (CTX + 1)
    pool0 = (torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).to(dt)
    qkv0 = (torch.randn(B, (HQ + 2 * HKV) * D, generator=g) * 0.5).to(dt)
    cos = torch.randn(B, 1, D // 2, generator=g).to(dt)
    sin = torch.randn(B, 1, D // 2, generator=g).to(dt)
    cos, sin = torch.cat([cos, cos], -1), torch.cat([sin, sin], -1)  # [B, 1, D] as the rotary producer hands out
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    perm = torch.randperm(rows, generator=g).int()
    table = perm[table.long()]  # scattered pool rows
    sel = table[:, CTX].contiguous()
    seq = torch.full((B,), CTX + 1, dtype=torch.int32)
    req = torch.arange(B, dtype=torch.int32)
    res0 = (torch.randn(B, HID, generator=g) * 0.5).to(dt)
    wn = (1 + 0.1 * torch.randn(HID, generator=g)).to(dt)
    scale = 1.0 / D**0.5

    pool = pool0.clone().to(DEV)
    qkv_in = qkv0.clone().to(DEV)
    res = res0.clone().to(DEV)
    d_cos, d_sin, d_table, d_sel, d_seq, d_req, d_wn = (t.to(DEV) for t in (cos, sin, table, sel, seq, req, wn))

    def step():
        qkv = qkv_in.clone()
        xq = qkv[:, : HQ * D].view(B, HQ, D)
        xkv = qkv[:, HQ * D:].view(B, 2 * HKV, D)
        rope_and_cache(xq, xkv, d_cos, d_sin, B, 1, d_sel, pool)
        o = K().flash_decoding(xq, pool[:, :HKV], pool[:, HKV:], scale, d_table, d_req, d_seq, CTX + 1)
        y, r = K().skip_rmsnorm(o.reshape(B, HQ * D), res.clone(), d_wn, 1e-6)
        return o, y, r

    eager = step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, captured):
        assert torch.equal(a, b)
    # oracle (fp32 arithmetic over the same bf16 values)
    xq = qkv0[:, : HQ * D].view(B, HQ, D).clone()
    xkv = qkv0[:, HQ * D:].view(B, 2 * HKV, D).clone()
    q_r, k_r = O.rope_emb_forward(xq, xkv[:, :HKV].contiguous(), cos, sin, B, 1)
    pool_ref = pool0.clone()
    new_rows = torch.cat([k_r, xkv[:, HKV:]], dim=1)
    O.update_kv_buffer(new_rows, sel, pool_ref)
    close(pool[sel.long().to(DEV)], pool_ref[sel.long()], 2e-2)
    o_ref = O.flash_decoding(q_r, pool_ref[:, :HKV], pool_ref[:, HKV:], scale, table, req, seq, CTX + 1)
    close(eager[0], o_ref, 2e-2)
    y_ref, r_ref = O.skip_rmsnorm(o_ref.reshape(B, HQ * D), res0.clone(), wn, 1e-6)
    close(eager[2], r_ref, 2e-2)
    close(eager[1], y_ref, 3e-2)


# ------------------------------------------------------------------------------------- #
# config 4: Llama-3-8B W8A8 widths at batch 32
# ------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("N,K_", [(6144, 4096), (4096, 4096), (28672, 4096), (14336, 4096), (4096, 14336)])
def test_config4_smoothquant_llama3_8b_widths(N, K_):
    """int8 codes, per-token scales and int32 accumulators bit-exact, fp16 output within the reference tolerance, for the
    fused q|k|v, o, fused gate|up, gate / up alone and down projections of Llama-3-8B (hidden 4096, MLP 14336)."""
    g = torch.Generator().manual_seed(N + K_)
    x = (torch.randn(32, K_, generator=g) * 0.5).half()
    qw = torch.randint(-127, 128, (N, K_), generator=g, dtype=torch.int32).to(torch.int8)
    ws = torch.rand(N, generator=g) * 0.01 + 0.002
    bias = (torch.randn(N, generator=g) * 0.1).half()
    out, acc, qa, a_scale = K().smoothquant_matmul(x.to(DEV), qw.to(DEV), ws.to(DEV), bias=bias.to(DEV), _return_int32=True)
    qa_ref, s_ref = O.quantize_activations_int8(x)
    assert torch.equal(qa.cpu(), qa_ref) and torch.equal(a_scale.cpu(), s_ref)
    rows = torch.randperm(N, generator=g)[:512].sort().values
    assert torch.equal(acc[:, rows.to(DEV)].cpu(), O.smoothquant_int32_acc(x, qw[rows])[0])
    ref = O.smoothquant_matmul(x, qw[rows], ws[rows], bias=bias[rows])
    close(out[:, rows.to(DEV)], ref, 1e-2)


# ------------------------------------------------------------------------------------- #
# RCCL: collectives of the decode step inside a hipGraph
# ------------------------------------------------------------------------------------- #
def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("LL_DIST_BACKEND", None)
    os.environ["LL_TP_FORCE_COLLECTIVE"] = "1"
    import torch.distributed as dist
    from lite_llama_amd.distributed import parallel_state as ps

    try:
        torch.cuda.set_device(0)
        ps.init_parallel(0, tp_size=1, dp_size=1, master_port=port)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and ps.collective_forced()
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(*a, **k):
            calls["n"] += 1
            return real(*a, **k)

        dist.all_reduce = counting
        from lite_llama_amd.executor import DecodeEngine
        from lite_llama_amd.model import CausalLM, tiny_geometry
        from lite_llama_amd.quantization import QuantConfig

        geo = tiny_geometry(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2, head_dim=64,
                            vocab_size=640, qkv_bias=True)
        quant = QuantConfig.int4_groupwise(128)
        model = CausalLM(geo, quant).init_synthetic(seed=9, quant=quant, device="cuda")
        g = torch.Generator().manual_seed(4)
        ids = torch.randint(0, 640, (2, 7), generator=g).cuda()
        outs = []
        for use_graph in (False, True):
            eng = DecodeEngine(model, max_batch=2, max_seq_len=32)
            first = eng.prefill(ids, torch.tensor([7, 5], device="cuda"))
            before = calls["n"]
            outs.append(eng.decode(first, 6, use_graph=use_graph).cpu())
            issued = calls["n"] - before
            # eager: 2 layers x 2 row-parallel projections x 6 steps; graph: warm-up + capture only
            assert issued == (24 if not use_graph else 8), (use_graph, issued)
        q.put((True, torch.equal(outs[0], outs[1]), outs[0].numpy()))
    except Exception as exc:  # pragma: no cover
        import traceback

        q.put((False, repr(exc) + traceback.format_exc()[-2000:], None))
    finally:
        ps.destroy_parallel()


def test_rccl_all_reduce_inside_captured_decode_step():
    """backend "nccl" (= RCCL) with a world of one rank: every row-parallel projection of the step issues a real
    all-reduce; the step is captured in a hipGraph (collectives included) and its replays produce the eager tokens."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    ok, same, toks = q.get(timeout=300)
    p.join(timeout=60)
    assert ok is True, same
    assert same is True, toks


def test_moe_model_with_interleaved_gate_up_equals_the_stacked_layout():
    """``CausalLM.compact_weights()`` on a Qwen3-MoE-shaped fp8 model pairs the gate|up rows of every expert (the fused
    epilogue route): same logits bit for bit as the stacked layout, route taken, and ``state_dict()`` / ``expand_weights()``
    give the checkpoint order back bit for bit."""
    import types
    from lite_llama_amd.model import CausalLM, SparseMoeBlock, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    geo = tiny_geometry(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=128,
                        vocab_size=512, qkv_bias=False, use_qk_norm=True, num_experts=16, num_experts_per_tok=4,
                        moe_intermediate_size=128)
    g = torch.Generator().manual_seed(3)
    m = CausalLM(geo)
    params = {k: ((1 + 0.1 * torch.randn(v.shape, generator=g)) if k.endswith("norm_weight") else 0.05 * torch.randn(v.shape, generator=g)).half()
              for k, v in m.state_dict().items()}
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.fp8_per_channel())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    B, CTX = 8, 40
    m.rotary_emb.ensure(CTX + 8, "cuda")
    rows = B * (CTX + 1)
    table = torch.arange(rows, dtype=torch.int32, device="cuda").view(B, CTX + 1)
    kv0 = [(torch.randn(rows, 4, 128, generator=g) * 0.5).half().cuda() for _ in range(2)]

    def run():
        kv = [k.clone() for k in kv0]
        info = types.SimpleNamespace(kv_buffer=kv, cur_select_index=table[:, CTX].contiguous(), b_req_tokens_table=table, b_start_loc=None,
                                     b_req_idx=torch.arange(B, dtype=torch.int32, device="cuda"),
                                     b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device="cuda"), max_actual_seq_len=CTX + 1)
        with torch.no_grad():
            return m(torch.arange(B, device="cuda").view(B, 1), torch.full((B, 1), CTX, device="cuda"), info)

    stacked = run()
    m.compact_weights()
    blocks = [b for b in m.modules() if isinstance(b, SparseMoeBlock)]
    assert blocks and all(getattr(b, "_gu_interleaved", False) for b in blocks) and m.is_compacted()
    assert torch.equal(run(), stacked)
    epoch = m.layout_epoch
    after = m.state_dict()  # exports the checkpoint order; the model -- and whatever was captured over it -- is not touched
    assert m.is_compacted() and m.layout_epoch == epoch and all(torch.equal(after[k], before[k]) for k in before)
    sub = blocks[0].state_dict()  # ... a submodule's export alike
    assert torch.equal(sub["experts.gate_up_proj"], before["layers.0.mlp.experts.gate_up_proj"])
    assert all(not p.requires_grad for p in m.parameters())
    assert torch.equal(run(), stacked)
    with pytest.raises(RuntimeError, match="expand_weights"):  # stacked rows must not be copied into interleaved storage
        blocks[0].load_state_dict(sub)
    m.load_state_dict(after, strict=True)  # the model-level load expands first
    assert not m.is_compacted() and m.layout_epoch == epoch + 1
    assert all(torch.equal(v, before[k]) for k, v in m.state_dict().items())
    assert torch.equal(run(), stacked)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unquantised_wide_mlp_and_lm_head_take_the_row_group_kernel(dtype, monkeypatch):
    """BASELINE config 2's route (round 5): an unquantised model whose fused gate|up and lm_head are wide enough runs them on
    the in-tree 16-bit row-group kernel -- gate|up + swiglu as ONE launch over row-interleaved weights, no library GEMM on the
    step -- and still matches the oracle; prefill (more rows than the kernel serves) reads the interleaved pairs in place."""
    import types
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from oracle.model import OracleModel
    import lite_llama_amd.quantization.methods as QM
    import lite_llama_amd.kernels.quantization as KQ
    import lite_llama_amd.model as M

    H, I, L, HQ, HKV, D, V = 256, 4096, 1, 2, 2, 128, 8192
    B, CTX = 4, 140
    g = torch.Generator().manual_seed(11)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV, head_dim=D, vocab_size=V,
                        qkv_bias=True)
    m = CausalLM(geo)
    params = {k: ((1 + 0.1 * torch.randn(v.shape, generator=g)) if k.endswith("norm_weight") else 0.03 * torch.randn(v.shape, generator=g)).half()
              for k, v in m.state_dict().items()}
    m.load_state_dict(params, strict=True)
    m = m.to("cuda").to(dtype)
    calls = {"rows": 0, "swiglu": 0}
    real = KQ.dense16_rows_linear

    def rows(x, w, bias=None, *, gate_up_swiglu=False):
        out = real(x, w, bias, gate_up_swiglu=gate_up_swiglu)
        calls["rows"] += int(out is not None)
        calls["swiglu"] += int(out is not None and gate_up_swiglu)
        return out

    monkeypatch.setattr(QM, "dense16_rows_linear", rows)
    monkeypatch.setattr(KQ, "dense16_rows_linear", rows)
    m.rotary_emb.ensure(CTX + 8, "cuda", dtype)
    rows_total = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows_total, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows_total, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
                                     b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
                                     b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", [k.clone().cuda().to(dtype) for k in kv_cpu]))
    assert calls == {"rows": 2, "swiglu": 1}, calls  # fused gate|up + swiglu, lm_head
    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps, rope_theta=geo.rope_theta)
    ref = om.forward(ids, pos, info_on("cpu", [k.clone() for k in kv_cpu]))
    tol = 6e-2 if dtype == torch.bfloat16 else 3e-2
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=tol, atol=tol)
    # prefill-shaped call of the same block: more rows than the kernel serves -> merged GEMM + in-place pair swiglu
    xp = (torch.randn(96, H, generator=g) * 0.5).to(dtype).cuda()
    mlp = m.layers[0].mlp
    with torch.no_grad():
        many = mlp(xp.view(1, 96, H)).view(96, H)
        few = torch.cat([mlp(xp[i: i + 32].view(1, 32, H)).view(32, H) for i in range(0, 96, 32)])
    torch.testing.assert_close(many.float(), few.float(), rtol=2e-2, atol=2e-2)
