"""Checkpoint ingestion, device tier: AutoAWQ / AutoGPTQ int4 checkpoints -- written to disk in their
published layouts by the numpy packers of oracle/w4_layouts.py (parity with the third-party libraries themselves is
UNPINNED, see that file) -- streamed through ``load_pretrained``: every linear's native parameters must equal the
oracle's conversion bit for bit, the fused K/V halves and the tensor-parallel cuts must land where the fp16 loader puts
them, and an activation-ordered (desc_act) checkpoint must compute the checkpoint's own dense meaning."""

import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from lite_llama_amd import weights
from lite_llama_amd.distributed import parallel_state as ps
from oracle import w4_layouts as WL

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, I, HQ, HKV, D, V, G = 256, 512, 4, 2, 64, 128, 64
LINEARS = (("self_attn.q_proj", HQ * D, H), ("self_attn.k_proj", HKV * D, H), ("self_attn.v_proj", HKV * D, H),
           ("self_attn.o_proj", H, HQ * D), ("mlp.gate_proj", I, H), ("mlp.up_proj", I, H), ("mlp.down_proj", H, I))


def _config(method, **extra):
    return {"model_type": "qwen2", "hidden_size": H, "intermediate_size": I, "num_hidden_layers": 1,
            "num_attention_heads": HQ, "num_key_value_heads": HKV, "vocab_size": V, "rms_norm_eps": 1e-6,
            "rope_theta": 10000.0, "tie_word_embeddings": False,
            "quantization_config": {"quant_method": method, "bits": 4, "group_size": G, **extra}}


def _write(directory, fmt, act_order=False, seed=5):
    """-> {module: (q [K, N], zeros [K/g, N], scales f16 [K/g, N], g_idx [K])} of what was written."""
    rng = np.random.default_rng(seed)
    state, truth = {}, {}
    f16 = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    state["model.embed_tokens.weight"] = f16((rng.standard_normal((V, H)) * 0.05).astype(np.float16))
    state["lm_head.weight"] = f16((rng.standard_normal((V, H)) * 0.05).astype(np.float16))
    state["model.norm.weight"] = f16((1 + 0.1 * rng.standard_normal(H)).astype(np.float16))
    p = "model.layers.0."
    state[p + "input_layernorm.weight"] = f16((1 + 0.1 * rng.standard_normal(H)).astype(np.float16))
    state[p + "post_attention_layernorm.weight"] = f16((1 + 0.1 * rng.standard_normal(H)).astype(np.float16))
    perms = {}
    for name, n, k in LINEARS:
        q = rng.integers(0, 16, (k, n), dtype=np.int64)
        z = rng.integers(1, 16, (k // G, n), dtype=np.int64)          # >= 1: representable in the v1 (z - 1) encoding
        s = (rng.random((k // G, n)) * 0.01 + 0.002).astype(np.float16)
        gi = np.arange(k) // G
        if act_order:
            src = "attn_in" if name.split(".")[-1] in ("q_proj", "k_proj", "v_proj") else "mlp_in" if "gate" in name or "up" in name else name
            if src not in perms:  # projections reading the same activations were quantised in the same channel order
                perms[src] = rng.permutation(k)
            gi = np.empty(k, dtype=np.int64)
            gi[perms[src]] = np.arange(k) // G
        if fmt == "awq":
            qw, qz = WL.awq_pack(q, z)
        else:
            qw, qz = WL.gptq_pack(q, z, v1=fmt == "gptq")
            state[p + name + ".g_idx"] = torch.from_numpy(gi.astype(np.int32))
        state[p + name + ".qweight"], state[p + name + ".qzeros"] = torch.from_numpy(qw.copy()), torch.from_numpy(qz.copy())
        state[p + name + ".scales"] = f16(s)
        if name.endswith(("q_proj", "k_proj", "v_proj")):
            state[p + name + ".bias"] = f16((rng.standard_normal(n) * 0.02).astype(np.float16))
        truth[name] = (q, z, s, gi)
    os.makedirs(directory, exist_ok=True)
    save_file(state, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})
    extra = {"version": "GEMM", "zero_point": True} if fmt == "awq" else {"desc_act": act_order, "checkpoint_format": fmt}
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump(_config("awq" if fmt == "awq" else "gptq", **extra), f)
    return truth


def _native(truth, name, fmt):
    """the oracle's conversion of one written linear -> (weight [N, K/8] int32, scale [N, K/g], zeros [N, K/g])"""
    q, z, s, _ = truth[name]
    if fmt == "awq":
        return WL.awq_to_native(*WL.awq_pack(q, z), s, G)
    return WL.gptq_to_native(*WL.gptq_pack(q, z, v1=fmt == "gptq"), s, G, v1=fmt == "gptq")


@pytest.fixture
def tp_state():
    yield
    ps._TP_WORLD_SIZE, ps._TP_RANK = 1, 0


def _layer_params(model, module):
    m = model.get_submodule("layers.0." + module)
    return m.weight.data.cpu().numpy(), m.weight_scale.data.cpu().numpy(), m.weight_zeros.data.cpu().numpy()


@pytest.mark.parametrize("fmt", ["awq", "gptq", "gptq_v2"])
def test_int4_checkpoint_loads_to_the_oracle_conversion(fmt, tmp_path):
    truth = _write(str(tmp_path), fmt)
    model = weights.load_pretrained(str(tmp_path), device=DEV, compact=False)
    for name, n, k in LINEARS:
        if name.endswith(("k_proj", "v_proj")):
            continue
        for got, want in zip(_layer_params(model, name), _native(truth, name, fmt)):
            assert got.shape == want.shape and np.array_equal(got, want), name
    # fused K/V: rows [K ; V] of every native tensor
    kv = _layer_params(model, "self_attn.kv_proj")
    for got, wk, wv in zip(kv, _native(truth, "self_attn.k_proj", fmt), _native(truth, "self_attn.v_proj", fmt)):
        assert np.array_equal(got, np.concatenate([wk, wv], axis=0))
    layer = model.layers[0]
    assert layer.self_attn.q_proj.quant.is_int4 and layer.self_attn.q_proj.quant.group_k == G
    assert layer.self_attn.kv_proj.bias.shape == (2 * HKV * D,)
    # the loaded layer computes the checkpoint's dense meaning
    x = (torch.randn(8, I, device=DEV) * 0.5).half()
    q, z, s, _ = truth["mlp.down_proj"]
    want = x.float().cpu().numpy() @ WL.dequant_kn(q, z, s, G)
    got = layer.mlp.down_proj(x).float().cpu().numpy()
    assert np.abs(got - want).max() <= 1e-2 * np.abs(want).max() + 1e-2


def test_loader_compacts_by_default_and_still_exports_the_checkpoint_layouts(tmp_path):
    """``load_pretrained`` leaves the configuration ``bench.py`` measures: one resident copy of the int4 words
    (parameters alias the decode engine's load-time layout), and ``state_dict()`` -- without touching the model -- exports
    exactly what the uncompacted load holds; both models compute the same rows."""
    _write(str(tmp_path), "awq")
    plain = weights.load_pretrained(str(tmp_path), device=DEV, compact=False)
    model = weights.load_pretrained(str(tmp_path), device=DEV)
    assert not plain.is_compacted()
    compacted = model.is_compacted()  # (layers whose shapes the decode engine's layout serves; this checkpoint's are small)
    want, got = plain.state_dict(), model.state_dict()
    assert model.is_compacted() == compacted and want.keys() == got.keys()
    for k in want:
        assert torch.equal(want[k], got[k]), k
    x = (torch.randn(8, I, device=DEV) * 0.5).half()
    assert torch.equal(model.layers[0].mlp.down_proj(x), plain.layers[0].mlp.down_proj(x))


def test_int4_checkpoint_tensor_parallel_cuts(tmp_path, tp_state):
    """Each rank's parameters are the slices of the TP = 1 parameters that the fp16 loader would give it: output rows for
    q / kv / gate / up (kv: [K_r ; V_r]), input columns (words / groups) for o / down."""
    truth = _write(str(tmp_path), "awq")
    full = {name: _native(truth, name, "awq") for name, _, _ in LINEARS}
    for rank in (0, 1):
        ps._TP_WORLD_SIZE, ps._TP_RANK = 2, rank
        model = weights.load_pretrained(str(tmp_path), device=DEV, compact=False)
        for name in ("self_attn.q_proj", "mlp.gate_proj", "mlp.up_proj"):
            for got, want in zip(_layer_params(model, name), full[name]):
                rows = want.shape[0] // 2
                assert np.array_equal(got, want[rank * rows:(rank + 1) * rows]), name
        for got, wk, wv in zip(_layer_params(model, "self_attn.kv_proj"), full["self_attn.k_proj"], full["self_attn.v_proj"]):
            r = wk.shape[0] // 2
            assert np.array_equal(got, np.concatenate([wk[rank * r:(rank + 1) * r], wv[rank * r:(rank + 1) * r]], axis=0))
        for name in ("self_attn.o_proj", "mlp.down_proj"):
            for got, want in zip(_layer_params(model, name), full[name]):
                cols = want.shape[1] // 2
                assert np.array_equal(got, want[:, rank * cols:(rank + 1) * cols]), name
        assert model.layers[0].self_attn.kv_proj.bias.shape == (HKV * D,)


def test_gptq_act_order_checkpoint(tmp_path, tp_state):
    """desc_act: the loader brings every linear into group order (bit-exact against the oracle conversion of the
    sorted tensor), remembers the permutation, and the layers -- incl. the fused q|k|v and gate|up launches, which need
    ONE input order -- compute (q - z[g_idx]) * s[g_idx]."""
    truth = _write(str(tmp_path), "gptq", act_order=True)
    model = weights.load_pretrained(str(tmp_path), device=DEV, compact=False)
    layer = model.layers[0]
    for name, n, k in LINEARS:
        if name.endswith(("k_proj", "v_proj")):
            continue
        q, z, s, gi = truth[name]
        perm = np.argsort(gi, kind="stable")
        m = model.get_submodule("layers.0." + name)
        assert np.array_equal(m.act_perm.cpu().numpy(), perm)
        want = WL.gptq_to_native(*WL.gptq_pack(q[perm], z, v1=True), s, G, v1=True)
        for got, w in zip(_layer_params(model, name), want):
            assert np.array_equal(got, w), name
    x = (torch.randn(8, H, device=DEV) * 0.5).half()

    def dense(name):
        q, z, s, gi = truth[name]
        return x.float().cpu().numpy() @ WL.dequant_kn_act_order(q, z, s, gi)

    def close(got, want):
        assert np.abs(got.float().cpu().numpy() - want).max() <= 1e-2 * np.abs(want).max() + 1e-2

    bias = lambda m: m.bias.float().cpu().numpy()
    close(layer.self_attn.q_proj(x), dense("self_attn.q_proj") + bias(layer.self_attn.q_proj))
    kvb = bias(layer.self_attn.kv_proj)
    close(layer.self_attn.kv_proj(x), np.concatenate([dense("self_attn.k_proj"), dense("self_attn.v_proj")], axis=1) + kvb)
    # the merged launches (decode path): q|k|v as one GEMM, gate|up + swiglu as one GEMM
    assert layer.self_attn._qkv.refresh() and layer.mlp._gate_up.refresh()
    qo, kvo = layer.self_attn._qkv(x)
    close(qo, dense("self_attn.q_proj") + bias(layer.self_attn.q_proj))
    gate, up = dense("mlp.gate_proj"), dense("mlp.up_proj")
    g16, u16 = gate.astype(np.float16).astype(np.float32), up.astype(np.float16).astype(np.float32)
    close(layer.mlp._gate_up.swiglu(x), g16 / (1 + np.exp(-g16)) * u16)
    # a row-parallel desc_act linear cannot be cut along its input channels
    ps._TP_WORLD_SIZE, ps._TP_RANK = 2, 0
    with pytest.raises(NotImplementedError, match="activation-ordered"):
        weights.load_pretrained(str(tmp_path), device=DEV, compact=False)


def test_fp16_checkpoint_quantised_at_load_time_runs_the_decode_engine(tmp_path):
    """--quantization int4 on an fp16 checkpoint (loader.py:139-147): streamed to the device, quantised layer by layer with
    the reference quantiser, then greedy decode through the hipGraph engine (graph == eager)."""
    from lite_llama_amd.executor import DecodeEngine

    rng = np.random.default_rng(9)
    t = lambda *shape, std=0.05: torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float16))
    p = "model.layers.0."
    state = {"model.embed_tokens.weight": t(V, H), "lm_head.weight": t(V, H), "model.norm.weight": 1 + t(H, std=0.1),
             p + "input_layernorm.weight": 1 + t(H, std=0.1), p + "post_attention_layernorm.weight": 1 + t(H, std=0.1)}
    for name, n, k in LINEARS:
        state[p + name + ".weight"] = t(n, k)
        if name.endswith(("q_proj", "k_proj", "v_proj")):
            state[p + name + ".bias"] = t(n, std=0.02)
    save_file(state, os.path.join(str(tmp_path), "model.safetensors"), metadata={"format": "pt"})
    cfg = _config("none")
    del cfg["quantization_config"]
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    model = weights.load_pretrained(str(tmp_path), device=DEV, quantization="int4", compact=False)
    down = model.layers[0].mlp.down_proj
    assert down.quant.is_int4 and down.weight.dtype == torch.int32 and down.weight.shape == (H, I // 8)
    outs = []
    for use_graph in (False, True):
        eng = DecodeEngine(model, max_batch=2, max_seq_len=32)
        ids = torch.randint(0, V, (2, 6), generator=torch.Generator().manual_seed(3)).to(DEV)
        first = eng.prefill(ids, torch.tensor([6, 4], device=DEV))
        outs.append(eng.decode(first, 5, use_graph=use_graph).cpu())
    assert torch.equal(outs[0], outs[1])
