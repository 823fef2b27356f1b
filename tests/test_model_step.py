"""Whole decode-step parity: call ORDER, KV-pool / table side effects and logits of the model-step
caller against the golden captured from the reference's own model code.

CPU tier: the oracle model vs the golden.  GPU tier: the HIP-backed model vs the golden + oracle.
"""

import types

import numpy as np
import pytest
import torch

from tests import _golden as G


# (family, quantisations pinned in its fixture); qwen2 from gen_golden_model.py, the rest from
# gen_golden_families.py
FAMILIES = {
    "qwen2": [None, "int4", "int8", "smoothquant", "fp8"],
    "llama": [None, "int4", "int8", "smoothquant", "fp8"],
    "qwen3": [None, "int4", "fp8"],
    "qwen3_moe": [None, "int8", "smoothquant", "fp8"],
}
CASES = [(f, q) for f, qs in FAMILIES.items() for q in qs]


def _load(family="qwen2"):
    d = np.load(G.GOLDEN_DIR + f"/model_step_{family}_tiny.npz")
    params = {k[len("param."):]: torch.from_numpy(d[k].copy()) for k in d.files if k.startswith("param.")}
    return d, params


def _extras(d):
    """(rope_theta, eps, qk_norm, moe) -- the qwen2 fixture predates these keys."""
    if "rope_theta" not in d.files:
        return 10000.0, 1e-6, False, None
    moe = tuple(int(x) for x in d["moe"])
    return float(d["rope_theta"]), float(d["eps"]), bool(int(d["qk_norm"])), (moe if moe[0] else None)


def _info(kv, table, sel, seq, start, max_len, dev="cpu"):
    return types.SimpleNamespace(kv_buffer=kv, cur_select_index=sel.to(dev), b_req_tokens_table=table,
                                 b_start_loc=start.to(dev) if start is not None else None,
                                 b_req_idx=torch.arange(len(seq), dtype=torch.int32, device=dev),
                                 b_seq_len=seq.to(dev), max_actual_seq_len=max_len)


def _run_steps(make_model, dev, quant=None, family="qwen2"):
    """prefill (fp16 always) + decode with ``quant``; returns (last prefill logits, decode logits, kv, table)."""
    d, params = _load(family)
    H, I, L, HQ, HKV, D, V = [int(x) for x in d["geometry"]]
    lens = d["lens"].tolist()
    B, LP = len(lens), max(lens)
    kv = [torch.zeros(64, 2 * HKV, D, dtype=torch.float16, device=dev) for _ in range(L)]
    table = torch.zeros(B, 16, dtype=torch.int32, device=dev)
    sel = torch.arange(B * LP, dtype=torch.int32)
    for i, n in enumerate(lens):
        table[i, :n] = sel[i * LP : i * LP + n].to(dev)
    info = _info(kv, table, sel, torch.tensor(lens, dtype=torch.int32), torch.arange(B, dtype=torch.int32) * LP, LP, dev)
    m16 = make_model(params, None, family)
    ids = torch.from_numpy(d["prompt_ids"]).to(dev)
    pos = torch.arange(LP, device=dev).unsqueeze(0).expand(B, LP).contiguous()
    logits = m16.forward(ids, pos, info)
    last = torch.stack([logits[i, n - 1] for i, n in enumerate(lens)])
    kv_prefill = [k.clone() for k in kv]
    # decode_alloc_kv_cache
    info.cur_select_index = torch.arange(B * LP, B * LP + B, dtype=torch.int32, device=dev)
    info.b_seq_len = info.b_seq_len + 1
    info.max_actual_seq_len += 1
    for i in range(B):
        table[i, int(info.b_seq_len[i]) - 1] = info.cur_select_index[i]
    tok = torch.from_numpy(d["first_tokens"]).to(dev)
    mq = make_model(params, quant, family) if quant else m16
    dl = mq.forward(tok.view(B, 1), torch.from_numpy(d["decode_positions"]).to(dev), info)
    return d, last, dl, kv_prefill, kv, table


def _oracle_model(params, quant, family="qwen2"):
    from oracle.model import OracleModel

    d = np.load(G.GOLDEN_DIR + f"/model_step_{family}_tiny.npz")
    H, I, L, HQ, HKV, D, V = [int(x) for x in d["geometry"]]
    theta, eps, qk_norm, moe = _extras(d)
    return OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=eps, rope_theta=theta,
                       quant=quant, qk_norm=qk_norm, moe=moe)


@pytest.mark.parametrize("family,quant", CASES)
def test_oracle_model_matches_reference_step(family, quant):
    d, last, dl, kvp, kvd, table = _run_steps(_oracle_model, "cpu", quant, family)
    lens = d["lens"].tolist()
    LP = max(lens)
    torch.testing.assert_close(last.float(), torch.from_numpy(d["logits_prefill_last"]).float(), rtol=2e-2, atol=2e-2)
    assert torch.equal(torch.argmax(last, -1), torch.from_numpy(d["first_tokens"]))
    # KV pool: rows of the valid prompt tokens and of the decode token (pad rows hold junk by design)
    valid = [i * LP + j for i, n in enumerate(lens) for j in range(n)]
    for li in range(len(kvp)):
        torch.testing.assert_close(kvp[li][valid].float(), torch.from_numpy(d[f"kv_prefill.{li}"])[valid].float(),
                                   rtol=2e-2, atol=2e-2)
    assert torch.equal(table, torch.from_numpy(d["table_after"]))
    key = "fp16" if quant is None else quant
    ref = torch.from_numpy(d[f"logits_decode.{key}"]).float()
    tol = 1e-1 if quant == "smoothquant" else 3e-2
    torch.testing.assert_close(dl.float(), ref, rtol=tol, atol=tol)
    if quant is None:
        dec_rows = [len(lens) * LP + i for i in range(len(lens))]
        for li in range(len(kvd)):
            torch.testing.assert_close(kvd[li][dec_rows].float(), torch.from_numpy(d[f"kv_decode.{li}"])[dec_rows].float(),
                                       rtol=2e-2, atol=2e-2)
        assert torch.equal(torch.argmax(dl[:, -1], -1), torch.argmax(ref[:, -1], -1))


def _hip_model(params, quant, family="qwen2"):
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    d = np.load(G.GOLDEN_DIR + f"/model_step_{family}_tiny.npz")
    H, I, L, HQ, HKV, D, V = [int(x) for x in d["geometry"]]
    theta, eps, qk_norm, moe = _extras(d)
    kw = {}
    if moe:
        kw = dict(num_experts=moe[0], num_experts_per_tok=moe[1], moe_intermediate_size=moe[2],
                  norm_topk_prob=bool(moe[3]))
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV,
                        head_dim=D, vocab_size=V, rope_theta=theta, rms_norm_eps=eps, qkv_bias=family == "qwen2",
                        use_qk_norm=qk_norm, **kw)
    m = CausalLM(geo)
    m.load_state_dict({k: v for k, v in params.items()}, strict=True)
    m = m.to("cuda")
    if quant:
        cfg = {"int4": QuantConfig.int4_groupwise(128), "int8": QuantConfig.int8_per_channel(),
               "smoothquant": QuantConfig.smoothquant_per_channel(), "fp8": QuantConfig.fp8_per_channel()}[quant]
        m.quantize_(cfg)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("family,quant", CASES)
def test_hip_model_matches_reference_step(family, quant):
    d, last, dl, kvp, kvd, table = _run_steps(_hip_model, "cuda", quant, family)
    lens = d["lens"].tolist()
    LP = max(lens)
    torch.testing.assert_close(last.float().cpu(), torch.from_numpy(d["logits_prefill_last"]).float(), rtol=3e-2, atol=3e-2)
    assert torch.equal(torch.argmax(last, -1).cpu(), torch.from_numpy(d["first_tokens"]))
    valid = [i * LP + j for i, n in enumerate(lens) for j in range(n)]
    for li in range(len(kvp)):
        torch.testing.assert_close(kvp[li][valid].float().cpu(), torch.from_numpy(d[f"kv_prefill.{li}"])[valid].float(),
                                   rtol=2e-2, atol=2e-2)
    assert torch.equal(table.cpu(), torch.from_numpy(d["table_after"]))
    key = "fp16" if quant is None else quant
    ref = torch.from_numpy(d[f"logits_decode.{key}"]).float()
    tol = 1e-1 if quant == "smoothquant" else 3e-2
    torch.testing.assert_close(dl.float().cpu(), ref, rtol=tol, atol=tol)
    if quant is None:
        assert torch.equal(torch.argmax(dl[:, -1], -1).cpu(), torch.argmax(ref[:, -1], -1))


@pytest.mark.gpu
def test_engine_graph_equals_eager_and_oracle():
    """hipGraph-captured decode == eager decode (byte-identical greedy tokens, like the reference's
    tests/compile/test_cuda_graph.py:50-70), and both follow the ORACLE model's greedy path: the oracle is
    teacher-forced with the engine's tokens and must rank the engine's choice first at every step (or within
    the fp16 noise of its own first choice: the logits of a tiny random model can tie to ~1e-2)."""
    from lite_llama_amd.executor import DecodeEngine

    d, params = _load()
    m = _hip_model(params, None)
    ids = torch.from_numpy(d["prompt_ids"]).cuda()
    lens = torch.from_numpy(d["lens"]).int().cuda()
    outs = []
    steps = 12
    for use_graph in (False, True):
        eng = DecodeEngine(m, max_batch=2, max_seq_len=64)
        first = eng.prefill(ids, lens)
        assert torch.equal(first.cpu(), torch.from_numpy(d["first_tokens"]))
        outs.append(eng.decode(first, steps, use_graph=use_graph).cpu())
    assert torch.equal(outs[0], outs[1])
    # several steps per graph launch: groups of 5 + a remainder of 2 one-step replays -- same tokens
    eng = DecodeEngine(m, max_batch=2, max_seq_len=64)
    fired = []
    grouped = eng.decode(eng.prefill(ids, lens), steps, use_graph=True, steps_per_graph=5, on_step=fired.append).cpu()
    assert torch.equal(grouped, outs[0]) and fired == [0, 5, 10, 11]
    # ---- oracle greedy path over the same prompt ----
    H, I, L, HQ, HKV, D, V = [int(x) for x in d["geometry"]]
    om = _oracle_model(params, None)
    lens_l = d["lens"].tolist()
    B, LP = len(lens_l), max(lens_l)
    kv = [torch.zeros(128, 2 * HKV, D, dtype=torch.float16) for _ in range(L)]
    table = torch.zeros(B, 64, dtype=torch.int32)
    sel = torch.arange(B * LP, dtype=torch.int32)
    for i, n in enumerate(lens_l):
        table[i, :n] = sel[i * LP: i * LP + n]
    info = _info(kv, table, sel, torch.tensor(lens_l, dtype=torch.int32), torch.arange(B, dtype=torch.int32) * LP, LP)
    pos = torch.arange(LP).unsqueeze(0).expand(B, LP).contiguous()
    logits = om.forward(torch.from_numpy(d["prompt_ids"]), pos, info)
    tok = torch.stack([logits[i, n - 1] for i, n in enumerate(lens_l)]).float().argmax(-1)
    assert torch.equal(tok, torch.from_numpy(d["first_tokens"]))
    seq = torch.tensor(lens_l, dtype=torch.int32)
    next_row = B * LP
    agree = 0
    for step in range(steps):
        info.cur_select_index = torch.arange(next_row, next_row + B, dtype=torch.int32)
        next_row += B
        seq = seq + 1
        info.b_seq_len = seq
        info.max_actual_seq_len = int(seq.max())
        for i in range(B):
            table[i, int(seq[i]) - 1] = info.cur_select_index[i]
        lg = om.forward(tok.view(B, 1), (seq - 1).view(B, 1).long(), info)[:, -1].float()
        mine = outs[0][:, step]
        top = lg.max(-1).values
        chosen = lg.gather(1, mine.view(B, 1)).squeeze(1)
        assert torch.all(top - chosen <= 2e-2), (step, top - chosen)  # the engine's token is the oracle's (near-)argmax
        agree += int((lg.argmax(-1) == mine).sum())
        tok = mine  # teacher-forced: both models see the same context at every step
    assert agree >= int(0.9 * steps * B), agree


@pytest.mark.gpu
def test_full_width_decode_layer_matches_oracle():
    """One decoder layer at the REAL Qwen2.5-7B widths (hidden 3584, intermediate 18944, 28/4 heads of
    128), int4 g128, one decode step for 4 sequences: exercises the fused [q|k|v] launch, the tile-group
    and stream-K GEMM plans, the interleaved gate/up + swiglu epilogue, position-indexed rope + KV
    scatter and flash-decoding at production shapes against the CPU oracle (vocab cut to 2048 to keep
    the oracle's lm_head cheap)."""
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from oracle.model import OracleModel

    H, I, L, HQ, HKV, D, V = 3584, 18944, 1, 28, 4, 128, 2048
    B, CTX = 4, 40
    g = torch.Generator().manual_seed(1234)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV,
                        head_dim=D, vocab_size=V, rope_theta=1000000.0, qkv_bias=True)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.1 * torch.randn(t.shape, generator=g)).half()
        elif name.endswith("bias"):
            params[name] = (0.01 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.02 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.int4_groupwise(128))
    m.rotary_emb.ensure(CTX + 8, "cuda")

    rows = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(
            kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
            b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
            b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    kv_gpu = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_gpu))
    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant="int4")
    kv_ref = [k.clone() for k in kv_cpu]
    ref = om.forward(ids, pos, info_on("cpu", kv_ref))
    new_rows = table[:, CTX].long()
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float().cpu(), kv_ref[0][new_rows].float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=3e-2, atol=3e-2)
    assert torch.equal(torch.argmax(got[:, -1], -1).cpu(), torch.argmax(ref[:, -1], -1))


@pytest.mark.gpu
def test_quantisers_on_the_device_equal_the_cpu_quantisers_bit_for_bit():
    """The load-time quantisers run wherever the weight lives (methods/*.py convert_from_fp16).  On the device torch turns
    ``x / 14.0`` into a multiplication by the reciprocal: scales 1 ulp off and ~1 nibble in 1400 one step off against the
    reference's CPU quantisers (found by the headline-shape layer test below: a 0.026 outlier in a new V row = one int4
    step x an activation of 3.6).  ``params._div_const`` keeps the division exact; codes, scales and zero points must be
    IDENTICAL on both devices and equal to the oracle's."""
    from lite_llama_amd.quantization import params as P
    from oracle import oracle as O

    g = torch.Generator().manual_seed(77)
    w = (torch.randn(1536, 3584, generator=g) * 0.02).half()
    for fn, ofn in ((lambda t: P.quantize_int4_groupwise(t, 128), lambda t: O.quantize_int4_groupwise(t, 128)),
                    (P.quantize_int8_per_channel, O.quantize_int8_per_channel),
                    (P.quantize_fp8_per_channel, O.quantize_fp8_per_channel),
                    (lambda t: P.quantize_int8_groupwise(t, 128), None)):
        on_cpu, on_gpu = fn(w), fn(w.cuda())
        for a, b in zip(on_cpu, on_gpu):
            assert torch.equal(a, b.cpu())
        if ofn is not None:
            for a, b in zip(on_cpu, ofn(w)):
                assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B,CTX,expect_partials", [(64, 549, True), (64, 1000, True), (128, 549, False), (64, 2048, True), (8, 2048, "long")])
def test_headline_shape_decode_layer_matches_oracle(B, CTX, expect_partials, monkeypatch):
    """The ASSEMBLED layer of the headline workload (round-2 review, "what's weak" 1): Qwen2.5-7B widths, int4 g128,
    batch 64 at a context of 549 / 1000 tokens -- the launch sequence bench.py times (fused q|k|v left as split-K
    partials -> one-launch attention over 5 / 8 partitions -> o partials -> add-and-normalise over partials ->
    gate|up + swiglu on 256-row tiles -> down partials -> norm) through ``model.py`` with ``partials_ok=True`` --
    against ``oracle/model.py`` on identical weights / K,V / tokens: logits at 3e-2, the new KV rows at 2e-2, greedy
    tokens.  Batch 128 takes the generic engines (M > 64: no pre-packed stream, no partials) through the same caller.
    The test also asserts WHICH route ran (a silent fall-back to the generic route would still pass the numbers)."""
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from oracle.model import OracleModel
    import lite_llama_amd.model as M
    import lite_llama_amd.kernels.quantization as Q

    H, I, L, HQ, HKV, D, V = 3584, 18944, 1, 28, 4, 128, 4096
    g = torch.Generator().manual_seed(4321 + B + CTX)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV,
                        head_dim=D, vocab_size=V, rope_theta=1000000.0, qkv_bias=True)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.1 * torch.randn(t.shape, generator=g)).half()
        elif name.endswith("bias"):
            params[name] = (0.01 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.02 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.int4_groupwise(128))
    m.rotary_emb.ensure(CTX + 8, "cuda")

    rows = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(
            kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
            b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
            b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    calls = {"attn_partials": 0, "norm_partials": 0, "gemm_partials": 0, "prepacked": 0}
    real_attn, real_norm = M.decode_attention_partials, M.skip_rmsnorm_partials
    real_part, real_pre = Q.w4a16_matmul_partials, Q.w4a16_matmul_prepacked

    def count(key, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            if out is not None:
                calls[key] += 1
            return out
        return wrapped

    monkeypatch.setattr(M, "decode_attention_partials", count("attn_partials", real_attn))
    monkeypatch.setattr(M, "skip_rmsnorm_partials", count("norm_partials", real_norm))
    import lite_llama_amd.quantization.methods as QM
    monkeypatch.setattr(QM, "w4a16_matmul_partials", count("gemm_partials", real_part))
    monkeypatch.setattr(QM, "w4a16_matmul_prepacked", count("prepacked", real_pre))

    kv_gpu = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_gpu))
    if expect_partials == "long":
        # 17 partitions at a SMALL batch (8 rows x 4 KV heads = 32 workgroups): the attention keeps a workgroup per partition +
        # the global merge (one row's context spread over many CUs) -- q|k|v is finished by the projection itself (pre-packed
        # stream, ordinary epilogue); o and down stay split-K partials into the two norms.  At batch 64 (round 6) the
        # one-workgroup form walks the 17 partitions with its 8 waves and takes the planes: the ordinary partial route above
        assert calls == {"attn_partials": 0, "norm_partials": 2, "gemm_partials": 2, "prepacked": 2}, calls
    elif expect_partials:
        # q|k|v, o and down as split-K partials; the attention and both norms consume them; gate|up on the pre-packed stream
        assert calls == {"attn_partials": 1, "norm_partials": 2, "gemm_partials": 3, "prepacked": 1}, calls
    else:
        assert calls == {"attn_partials": 0, "norm_partials": 0, "gemm_partials": 0, "prepacked": 0}, calls
    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant="int4")
    kv_ref = [k.clone() for k in kv_cpu]
    ref = om.forward(ids, pos, info_on("cpu", kv_ref))
    new_rows = table[:, CTX].long()
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float().cpu(), kv_ref[0][new_rows].float(), rtol=2e-2, atol=2e-2)
    old_rows = table[:, :CTX].reshape(-1)[:: 97].long()  # the context rows are untouched
    assert torch.equal(kv_gpu[0][old_rows.cuda()].cpu(), kv_cpu[0][old_rows])
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=3e-2, atol=3e-2)
    # greedy tokens: equal wherever the oracle's top two logits are further apart than the tolerance
    top2 = ref[:, -1].float().topk(2, -1).values
    decided = (top2[:, 0] - top2[:, 1]) > 6e-2
    assert decided.sum() >= B // 2
    assert torch.equal(torch.argmax(got[:, -1], -1).cpu()[decided], torch.argmax(ref[:, -1], -1)[decided])


@pytest.mark.gpu
@pytest.mark.parametrize("quant,CTX,via_partials", [("fp8", 300, True), (None, 300, True), ("int4", 300, True), ("fp8", 1500, False)])
def test_qwen3_shape_decode_layer_fuses_the_head_norm_and_matches_oracle(quant, CTX, via_partials, monkeypatch):
    """Qwen3 attention geometry (32 / 4 heads of 128, q_norm / k_norm, no projection bias -- models/qwen3.py) at decode: the
    per-head norms run INSIDE the one-launch attention (no norm / rope / cat / scatter launches), fed by the q|k|v split-K
    partials while the context fits eight partitions -- against oracle/model.py with qk_norm on identical weights, and
    against the unfused route (LL_NO_FUSED_QK_NORM: two norm launches, rope, KV scatter, flash_decoding)."""
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from oracle.model import OracleModel
    import lite_llama_amd.model as M

    H, I, L, HQ, HKV, D, V = 2048, 2048, 1, 32, 4, 128, 1024
    B = 8
    g = torch.Generator().manual_seed(99 + CTX)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV, head_dim=D,
                        vocab_size=V, rope_theta=1000000.0, rms_norm_eps=1e-6, qkv_bias=False, use_qk_norm=True)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.2 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.02 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    if quant:
        m.quantize_({"int4": QuantConfig.int4_groupwise(128), "fp8": QuantConfig.fp8_per_channel()}[quant])
    m.rotary_emb.ensure(CTX + 8, "cuda")

    rows = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(
            kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
            b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
            b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    calls = {"attn_partials": 0, "attn": 0, "norm": 0}

    def count(key, fn, need_norm):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            if out is not None and (not need_norm or k.get("qk_norm") is not None):
                calls[key] += 1
            return out
        return wrapped

    monkeypatch.setattr(M, "decode_attention_partials", count("attn_partials", M.decode_attention_partials, True))
    monkeypatch.setattr(M, "decode_attention", count("attn", M.decode_attention, True))
    real_norm = M.skip_rmsnorm

    def norm_counted(x, r, w, eps):
        calls["norm"] += int(w.numel() == D)     # the head norms (weights of one head's width)
        return real_norm(x, r, w, eps)

    monkeypatch.setattr(M, "skip_rmsnorm", norm_counted)
    kv_gpu = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_gpu))
    assert calls == {"attn_partials": int(via_partials), "attn": int(not via_partials), "norm": 0}, calls

    monkeypatch.setenv("LL_NO_FUSED_QK_NORM", "1")
    monkeypatch.setenv("LL_NO_QKV_PARTIALS", "1")
    kv_two = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        two = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_two))
    assert calls["norm"] == 2
    new_rows = table[:, CTX].long()
    # (the two routes add the projection's split-K planes in different orders: close, not equal -- the bit-for-bit statement
    # on identical q | k | v is tests/test_kernels_gpu.py::test_decode_attention_with_head_norm_equals_...)
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float(), kv_two[0][new_rows.cuda()].float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(got.float(), two.float(), rtol=2e-2, atol=2e-2)

    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant=quant, qk_norm=True)
    kv_ref = [k.clone() for k in kv_cpu]
    ref = om.forward(ids, pos, info_on("cpu", kv_ref))
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float().cpu(), kv_ref[0][new_rows].float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=3e-2, atol=3e-2)


@pytest.mark.gpu
def test_smoothquant_shape_decode_layer_route_and_oracle(monkeypatch):
    """Llama-3 attention geometry (heads of 128, GQA 4, no bias), SmoothQuant W8A8, one decode step: the block runs
    quantiser-in-norm -> int8 q|k|v planes -> one-launch attention over the int32 planes -> o planes -> norm + quantiser ->
    gate|up planes -> finish + swiglu -> down planes (no finish launch anywhere) -- route asserted, logits against
    oracle/model.py (smoothquant) and against the unfused route (LL_NO_Q8_FUSION / LL_W8A8_NO_PARTIALS)."""
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from lite_llama_amd.kernels.norm_act import Int8Rows, ScaledInt32Partials
    from oracle.model import OracleModel
    import lite_llama_amd.model as M

    H, I, L, HQ, HKV, D, V = 2048, 4096, 1, 16, 4, 128, 1024
    B, CTX = 8, 300
    g = torch.Generator().manual_seed(77)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV, head_dim=D,
                        vocab_size=V, rope_theta=500000.0, rms_norm_eps=1e-5, qkv_bias=False)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.1 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.02 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.smoothquant_per_channel())
    m.rotary_emb.ensure(CTX + 8, "cuda")
    rows = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(
            kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
            b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
            b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    calls = {"attn_int32": 0, "norm_q8_planes": 0, "norm_q8_rows": 0}
    real_attn, real_q8 = M.decode_attention_partials, M.skip_rmsnorm_q8

    def attn(parts, *a, **k):
        out = real_attn(parts, *a, **k)
        calls["attn_int32"] += int(out is not None and isinstance(parts, ScaledInt32Partials))
        return out

    def q8(X, *a, **k):
        out = real_q8(X, *a, **k)
        calls["norm_q8_planes"] += int(isinstance(X, ScaledInt32Partials))
        calls["norm_q8_rows"] += int(isinstance(out[0], Int8Rows))
        return out

    monkeypatch.setattr(M, "decode_attention_partials", attn)
    monkeypatch.setattr(M, "skip_rmsnorm_q8", q8)
    kv_gpu = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_gpu))
    # input norm (fp16 in, int8 rows out), post-attention norm (o planes in, rows out), final norm (down planes in, fp16 out)
    assert calls == {"attn_int32": 1, "norm_q8_planes": 2, "norm_q8_rows": 2}, calls

    monkeypatch.setattr(M, "_Q8_FUSION", False)
    monkeypatch.setenv("LL_W8A8_NO_PARTIALS", "1")
    kv_two = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        two = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_two))
    assert torch.equal(got, two) and torch.equal(kv_gpu[0], kv_two[0])   # integer sums are exact: the two routes agree bit for bit

    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant="smoothquant")
    kv_ref = [k.clone() for k in kv_cpu]
    ref = om.forward(ids, pos, info_on("cpu", kv_ref))
    new_rows = table[:, CTX].long()
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float().cpu(), kv_ref[0][new_rows].float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=1e-1, atol=1e-1)


@pytest.mark.gpu
def test_prefill_then_decode_at_batch_64_follows_the_oracle(monkeypatch):
    """The reference's protocol end to end at batch 64 (benchmarks/common.py:100-137: prefill, then decode): int4 model with
    widths on the M-tiled engine's grid, 64 prompts of 96..128 tokens through ``DecodeEngine.prefill`` (every projection a
    > 64-row call: route asserted), then 6 captured decode steps.  The oracle model, teacher-forced with the engine's tokens,
    must rank the engine's choice first (or within fp16 noise of its own first choice) at the prefill step and at every
    decode step, and the K/V rows the prefill wrote must match the oracle's."""
    from lite_llama_amd.executor import DecodeEngine
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from oracle.model import OracleModel
    import lite_llama_amd.quantization.methods as QM

    H, I, L, HQ, HKV, D, V = 512, 1280, 2, 4, 2, 128, 1024
    B, LP, STEPS = 64, 128, 6
    g = torch.Generator().manual_seed(31)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV, head_dim=D,
                        vocab_size=V, rope_theta=10000.0, rms_norm_eps=1e-6, qkv_bias=True)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.1 * torch.randn(t.shape, generator=g)).half()
        elif name.endswith(".bias"):
            params[name] = (0.1 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.05 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.int4_groupwise(128))
    m.compact_weights()
    lens = torch.randint(96, LP + 1, (B,), generator=g).int()
    lens[0] = LP
    ids = torch.randint(0, V, (B, LP), generator=g)
    calls = {"rows": 0, "generic": 0}
    real_rows, real_gen = QM.w4a16_matmul_prepacked_rows, QM.w4a16_matmul

    def rows(*a, **k):
        calls["rows"] += 1
        return real_rows(*a, **k)

    def gen(*a, **k):
        calls["generic"] += 1
        return real_gen(*a, **k)

    monkeypatch.setattr(QM, "w4a16_matmul_prepacked_rows", rows)
    monkeypatch.setattr(QM, "w4a16_matmul", gen)
    eng = DecodeEngine(m, max_batch=B, max_seq_len=LP + STEPS + 8)
    first = eng.prefill(ids.cuda(), lens.cuda())
    assert calls["generic"] == 0 and calls["rows"] == 4 * L, calls  # q|k|v, o, gate|up (+ swiglu), down per layer
    toks = eng.decode(first, STEPS, use_graph=True).cpu()
    kv_gpu = [k.clone().cpu() for k in eng.info.kv_buffer]

    # ---- the oracle over the same prompt ----
    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant="int4")
    rows_total = B * (LP + STEPS + 8)
    kv = [torch.zeros(rows_total, 2 * HKV, D, dtype=torch.float16) for _ in range(L)]
    table = torch.zeros(B, LP + STEPS + 8, dtype=torch.int32)
    sel = torch.arange(B * LP, dtype=torch.int32)
    lens_l = lens.tolist()
    for i, n in enumerate(lens_l):
        table[i, :n] = sel[i * LP: i * LP + n]
    info = _info(kv, table, sel, lens.clone(), torch.arange(B, dtype=torch.int32) * LP, LP)
    pos = torch.arange(LP).unsqueeze(0).expand(B, LP).contiguous()
    logits = om.forward(ids, pos, info)
    lg = torch.stack([logits[i, n - 1] for i, n in enumerate(lens_l)]).float()
    chosen = lg.gather(1, first.cpu().view(B, 1)).squeeze(1)
    assert torch.all(lg.max(-1).values - chosen <= 3e-2)
    valid = torch.cat([sel[i * LP: i * LP + n] for i, n in enumerate(lens_l)]).long()
    for li in range(L):  # K/V rows of the valid prompt tokens (pad rows hold junk on both sides)
        torch.testing.assert_close(kv_gpu[li][valid].float(), kv[li][valid].float(), rtol=3e-2, atol=3e-2)
    seq = lens.clone()
    next_row = B * LP
    tok = first.cpu()
    agree = 0
    for step in range(STEPS):
        info.cur_select_index = torch.arange(next_row, next_row + B, dtype=torch.int32)
        next_row += B
        seq = seq + 1
        info.b_seq_len = seq
        info.max_actual_seq_len = int(seq.max())
        for i in range(B):
            table[i, int(seq[i]) - 1] = info.cur_select_index[i]
        lg = om.forward(tok.view(B, 1), (seq - 1).view(B, 1).long(), info)[:, -1].float()
        mine = toks[:, step]
        chosen = lg.gather(1, mine.view(B, 1)).squeeze(1)
        assert torch.all(lg.max(-1).values - chosen <= 3e-2), (step, (lg.max(-1).values - chosen).max())
        agree += int((lg.argmax(-1) == mine).sum())
        tok = mine
    assert agree >= int(0.9 * STEPS * B), agree


@pytest.mark.gpu
def test_smoothquant_with_bias_and_small_heads_keeps_the_finished_projection(monkeypatch):
    """Qwen2-style attention (q|k|v bias, heads of 64) under SmoothQuant: the one-launch attention does not take int32
    planes for this geometry, so the step must not leave q|k|v as planes at all (planes GEMM + a torch
    finish that added the bias after a first rounding) -- the q|k|v values equal ``smoothquant_matmul`` bit for bit and the
    logits match the oracle."""
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig
    from lite_llama_amd.kernels.norm_act import ScaledInt32Partials
    from oracle.model import OracleModel
    import lite_llama_amd.model as M

    H, I, L, HQ, HKV, D, V = 1024, 2048, 1, 16, 4, 64, 512
    B, CTX = 8, 200
    g = torch.Generator().manual_seed(5)
    geo = tiny_geometry(hidden_size=H, intermediate_size=I, num_layers=L, num_heads=HQ, num_kv_heads=HKV, head_dim=D,
                        vocab_size=V, rope_theta=10000.0, rms_norm_eps=1e-6, qkv_bias=True)
    m = CausalLM(geo)
    params = {}
    for name, t in m.state_dict().items():
        if name.endswith("norm_weight") or name.endswith("layernorm_weight"):
            params[name] = (1 + 0.1 * torch.randn(t.shape, generator=g)).half()
        elif name.endswith(".bias"):
            params[name] = (0.3 * torch.randn(t.shape, generator=g)).half()
        else:
            params[name] = (0.02 * torch.randn(t.shape, generator=g)).half()
    m.load_state_dict(params, strict=True)
    m = m.to("cuda")
    m.quantize_(QuantConfig.smoothquant_per_channel())
    m.rotary_emb.ensure(CTX + 8, "cuda")
    rows = B * (CTX + 1)
    kv_cpu = [(torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half() for _ in range(L)]
    table = torch.arange(rows, dtype=torch.int32).view(B, CTX + 1)
    ids = torch.randint(0, V, (B, 1), generator=g)
    pos = torch.full((B, 1), CTX)

    def info_on(dev, kv):
        return types.SimpleNamespace(
            kv_buffer=kv, cur_select_index=table[:, CTX].contiguous().to(dev), b_req_tokens_table=table.clone().to(dev),
            b_start_loc=None, b_req_idx=torch.arange(B, dtype=torch.int32, device=dev),
            b_seq_len=torch.full((B,), CTX + 1, dtype=torch.int32, device=dev), max_actual_seq_len=CTX + 1)

    seen = {"int32_planes_into_attention": 0}
    real_attn = M.decode_attention_partials

    def attn(parts, *a, **k):
        seen["int32_planes_into_attention"] += int(isinstance(parts, ScaledInt32Partials))
        return real_attn(parts, *a, **k)

    monkeypatch.setattr(M, "decode_attention_partials", attn)
    kv_gpu = [k.clone().cuda() for k in kv_cpu]
    with torch.no_grad():
        got = m(ids.cuda(), pos.cuda(), info_on("cuda", kv_gpu))
    assert seen["int32_planes_into_attention"] == 0
    om = OracleModel({k: v.clone() for k, v in params.items()}, H, I, L, HQ, HKV, D, V, eps=geo.rms_norm_eps,
                     rope_theta=geo.rope_theta, quant="smoothquant")
    kv_ref = [k.clone() for k in kv_cpu]
    ref = om.forward(ids, pos, info_on("cpu", kv_ref))
    new_rows = table[:, CTX].long()
    torch.testing.assert_close(kv_gpu[0][new_rows.cuda()].float().cpu(), kv_ref[0][new_rows].float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=1e-1, atol=1e-1)

    # the fallback value itself (a geometry check that fails AFTER the planes were made): one rounding, bias in fp32
    from lite_llama_amd.kernels.quantization import smoothquant_matmul, smoothquant_matmul_partials
    x = (torch.randn(B, H, generator=g) * 0.5).half().cuda()
    lin = m.layers[0].self_attn.q_proj
    planes = smoothquant_matmul_partials(x, lin.weight, lin.weight_scale_inv, bias=None)
    one = ScaledInt32Partials(planes.parts, planes.shape, planes.a_scale, planes.w_scale, bias=lin.bias).materialise()
    assert torch.equal(one, smoothquant_matmul(x, lin.weight, lin.weight_scale_inv, bias=lin.bias))


@pytest.mark.gpu
def test_engine_sampling_path_graph_equals_eager():
    """Non-greedy decode through the sampler kernels inside the captured step: with a vanishing top_p the
    nucleus is the single most probable token, so the stochastic path must reproduce greedy decoding
    exactly (whatever the uniform numbers), in graph and in eager mode; a repetition penalty changes
    the tokens but still has to agree between graph and eager."""
    from lite_llama_amd.executor import DecodeEngine

    d, params = _load()
    m = _hip_model(params, None)
    ids = torch.from_numpy(d["prompt_ids"]).cuda()
    lens = torch.from_numpy(d["lens"]).int().cuda()

    def run(sampling, use_graph):
        eng = DecodeEngine(m, max_batch=2, max_seq_len=64)
        return eng.decode(eng.prefill(ids, lens), 10, use_graph=use_graph, sampling=sampling).cpu()

    greedy = run(None, True)
    top1 = types.SimpleNamespace(temperature=0.7, top_p=1e-6, repetition_penalty=1.0)
    assert torch.equal(run(top1, True), greedy) and torch.equal(run(top1, False), greedy)
    pen = types.SimpleNamespace(temperature=0.7, top_p=1e-6, repetition_penalty=1.3)
    a, b = run(pen, True), run(pen, False)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_engine_penalised_decode_follows_the_reference_loop():
    """The generation loop of the reference (engine/llm_engine.py:150-176) restated step by step: the prefill step samples
    the first token with the same sampler, and from then on the repetition penalty sees EVERY generated token --
    the prefill-sampled one included (GeneratedSpan over tokens[:, :cur_pos]).  With a vanishing top_p the draw is the
    arg-max of the penalised logits, so the engine (graph and eager) must reproduce the restated loop token by token."""
    from lite_llama_amd.executor import DecodeEngine
    from oracle import oracle as O

    d, params = _load()
    m = _hip_model(params, None)
    ids = torch.from_numpy(d["prompt_ids"]).cuda()
    lens = torch.from_numpy(d["lens"]).int().cuda()
    sp = types.SimpleNamespace(temperature=0.7, top_p=1e-6, repetition_penalty=1.6)
    steps = 10
    runs = []
    for use_graph in (True, False):
        eng = DecodeEngine(m, max_batch=2, max_seq_len=64)
        first = eng.prefill(ids, lens, sampling=sp)
        runs.append((first.cpu(), eng.decode(first, steps, use_graph=use_graph, sampling=sp).cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    first, toks = runs[0]
    assert torch.equal(first, torch.from_numpy(d["first_tokens"]))  # nothing generated yet: the penalty cannot act
    # ---- the reference loop, restated over the engine's own eager logits (teacher-forced with its tokens) ----
    eng = DecodeEngine(m, max_batch=2, max_seq_len=64)
    f2 = eng.prefill(ids, lens)
    logits_seen = []
    orig = m.forward

    def spy(*a, **k):
        out = orig(*a, **k)
        logits_seen.append(out[:, -1].detach().float().cpu())
        return out

    m.forward = spy
    try:
        # feed the engine's penalised tokens back so that the contexts are identical
        eng._sampling = sp
        eng._begin_decode(f2, steps)
        for i in range(steps):
            eng.info.max_actual_seq_len = int(eng.info.max_actual_seq_len)
            lg = m(eng._input_ids, eng._positions, eng.info)[:, -1]
            eng._advance(toks[:, i].cuda())
            eng.info.max_actual_seq_len += 1
    finally:
        m.forward = orig
    B = toks.shape[0]
    generated = first.view(B, 1)
    for i in range(steps):
        lg = logits_seen[i].half()
        mask = torch.ones_like(generated, dtype=torch.bool)
        pen = O.apply_repetition_penalty(lg, generated, mask, sp.repetition_penalty)
        want = pen.float().argmax(-1)
        top2 = pen.float().topk(2, -1).values
        decided = (top2[:, 0] - top2[:, 1]) > 2e-2  # skip rows whose top two tie within fp16 noise
        assert torch.equal(toks[decided, i], want[decided]), (i, toks[:, i], want)
        generated = torch.cat([generated, toks[:, i].view(B, 1)], dim=1)
    # the penalty on the prefill-sampled token matters in this run: dropping it from the span changes the path
    lg0 = logits_seen[0].half()
    without = lg0.float().argmax(-1)
    with_pen = O.apply_repetition_penalty(lg0, first.view(B, 1), torch.ones(B, 1, dtype=torch.bool), sp.repetition_penalty).float().argmax(-1)
    assert torch.equal(toks[:, 0], with_pen) or not torch.equal(without, with_pen)
