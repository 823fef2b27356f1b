"""Checkpoint ingestion, CPU tier: the key -> parameter placement, the tensor-parallel cuts and the
streaming loader against what the REFERENCE loader produced from the same checkpoint files
(tests/golden/gen_golden_loader.py: parameters of ``model.load_weights(hf_weights_iterator(dir))`` at TP = 1 and on both
ranks of TP = 2, recorded as dtype / shape / SHA-256), and the coverage accounting with the error behaviour of the
reference's tests/models/test_weight_mapping.py."""

import hashlib
import json
import os

import pytest
import torch
import torch.nn as nn

from lite_llama_amd import weights
from lite_llama_amd.distributed import parallel_state as ps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "loader_expected.json")) as f:
    EXPECTED = json.load(f)


# ------------------------------------------------------------------------------------- #
# placement tables == the reference's answers
# ------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("key", sorted(EXPECTED["translate_text_key"]))
def test_translate_text_key_matches_reference(key):
    want_name, want_shape, want_first = EXPECTED["translate_text_key"][key]
    name, region = weights.translate_text_key(key)
    assert name == want_name
    probe = (torch.arange(12 * 8 * 3, dtype=torch.float32).view(12, 8, 3) if ".experts." in key
             else torch.arange(8 * 3, dtype=torch.float32).view(8, 3))
    view = region(probe)
    assert list(view.shape) == want_shape and float(view.reshape(-1)[0]) == want_first


def test_shard_dim_matches_reference():
    for name, want in EXPECTED["shard_dim"].items():
        assert weights.shard_dim(name) == want, name


def test_k_and_v_fill_opposite_halves_and_experts_their_own_slice():
    p = torch.zeros(8, 3)
    weights.translate_text_key("layers.0.self_attn.k_proj.weight")[1](p).fill_(1)
    weights.translate_text_key("layers.0.self_attn.v_proj.weight")[1](p).fill_(2)
    assert torch.equal(p[:4], torch.ones(4, 3)) and torch.equal(p[4:], torch.full((4, 3), 2.0))
    s = torch.zeros(3, 4, 5)
    weights.translate_text_key("layers.0.mlp.experts.1.gate_proj.weight")[1](s).fill_(1)
    weights.translate_text_key("layers.0.mlp.experts.1.up_proj.weight")[1](s).fill_(2)
    assert torch.equal(s[1, :2], torch.ones(2, 5)) and torch.equal(s[1, 2:], torch.full((2, 5), 2.0))
    assert s[0].abs().sum() == 0 and s[2].abs().sum() == 0
    assert weights.strip_prefix("model.layers.0", "model.") == "layers.0"
    assert weights.strip_prefix("visual.blocks.0", "model.") is None
    assert weights.strip_prefix("lm_head.weight", "") == "lm_head.weight"


# ------------------------------------------------------------------------------------- #
# coverage accounting: the reference's error behaviour
# ------------------------------------------------------------------------------------- #
class _Three(nn.Module):
    def __init__(self):
        super().__init__()
        self.plain = nn.Parameter(torch.zeros(2, 3))
        self.fused = nn.Parameter(torch.zeros(4, 3))
        self.mirror = nn.Parameter(torch.zeros(2, 3))


def _translate(key):
    table = {"plain": ("plain", weights.whole), "fused_low": ("fused", weights.half(0)),
             "fused_high": ("fused", weights.half(1)), "mirror": ("mirror", weights.whole), "ignore_me": None}
    return table.get(key, (key, weights.whole))


def _stream(drop=()):
    full = {"plain": torch.ones(2, 3), "fused_low": torch.full((2, 3), 4.0), "fused_high": torch.full((2, 3), 5.0),
            "mirror": torch.full((2, 3), 7.0), "ignore_me": torch.zeros(1)}
    return [(k, v) for k, v in full.items() if k not in drop]


def test_load_weights_accounting():
    m = _Three()
    weights.load_weights(m, _stream(), _translate)
    assert torch.equal(m.plain, torch.ones(2, 3)) and torch.equal(m.fused[:2], torch.full((2, 3), 4.0))
    assert torch.equal(m.fused[2:], torch.full((2, 3), 5.0)) and torch.equal(m.mirror, torch.full((2, 3), 7.0))
    with pytest.raises(ValueError, match=r"never written.*plain"):
        weights.load_weights(_Three(), _stream(drop=("plain",)), _translate)
    with pytest.raises(ValueError, match=r"partially written.*fused"):
        weights.load_weights(_Three(), _stream(drop=("fused_high",)), _translate)
    with pytest.raises(ValueError, match=r"plain \(12 of 6 elements\)"):
        weights.load_weights(_Three(), [*_stream(), ("plain", torch.zeros(2, 3))], _translate)
    with pytest.raises(ValueError, match=r"shape \(3, 3\) but 'plain' expects \(2, 3\)"):
        weights.load_weights(_Three(), [("plain", torch.zeros(3, 3))], _translate)
    with pytest.raises(ValueError, match="unknown parameter 'typo'"):
        weights.load_weights(_Three(), [("typo", torch.zeros(2, 3))], _translate)
    m = _Three()
    weights.load_weights(m, _stream(drop=("mirror",)), _translate, tied={"mirror": "plain"})
    assert torch.equal(m.mirror, m.plain)
    m = _Three()
    weights.load_weights(m, _stream(), _translate, tied={"mirror": "plain"})  # a tie never overrides a shipped tensor
    assert torch.equal(m.mirror, torch.full((2, 3), 7.0))


def test_int4_keys_are_unknown_parameters_without_the_extension():
    """As in the reference (weights.py:266-268): a qweight key has no parameter to land in."""
    with pytest.raises(ValueError, match="unknown parameter 'plain.qweight'"):
        weights.load_weights(_Three(), [("plain.qweight", torch.zeros(2, 3, dtype=torch.int32))], _translate)


def test_weight_file_discovery(tmp_path):
    (tmp_path / "model.safetensors").touch()
    (tmp_path / "pytorch_model.bin").touch()
    assert [p.name for p in weights.hf_weight_files(tmp_path)] == ["model.safetensors"]
    (tmp_path / "model.safetensors").unlink()
    assert [p.name for p in weights.hf_weight_files(tmp_path)] == ["pytorch_model.bin"]
    (tmp_path / "pytorch_model.bin").unlink()
    with pytest.raises(FileNotFoundError, match="no \\*.safetensors or \\*.bin"):
        weights.hf_weight_files(tmp_path)


# ------------------------------------------------------------------------------------- #
# whole checkpoints: every parameter bit-identical to the reference loader's, TP = 1 and both ranks of TP = 2
# ------------------------------------------------------------------------------------- #
def _digest(t):
    raw = t.detach().contiguous().view(torch.uint8).numpy().tobytes()
    return [str(t.dtype).replace("torch.", ""), list(t.shape), hashlib.sha256(raw).hexdigest()]


@pytest.fixture
def tp_state():
    yield
    ps._TP_WORLD_SIZE, ps._TP_RANK = 1, 0


@pytest.mark.parametrize("case", sorted(EXPECTED["cases"]))
@pytest.mark.parametrize("world,rank", [(1, 0), (2, 0), (2, 1)])
def test_checkpoint_loads_bit_identical_to_reference(case, world, rank, tp_state):
    ps._TP_WORLD_SIZE, ps._TP_RANK = world, rank  # loading needs the grid position only, no process group
    model = weights.load_pretrained(os.path.join(GOLDEN, "ckpt", case), device="cpu")
    got = {name: _digest(p.data) for name, p in model.named_parameters()}
    want = EXPECTED["cases"][case][f"tp{world}_rank{rank}"]
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name


def test_fp8_checkpoint_widened_on_the_way_in_equals_block_dequant(tp_state):
    """dequantize_fp8=True (an fp16 model fed from a block-fp8 checkpoint, weight_utils.py:58-71): W = w8 * s[i//128, j//128]
    in fp32, rounded once to fp16; the scale tables are consumed."""
    from safetensors import safe_open

    d = os.path.join(GOLDEN, "ckpt", "qwen3_moe_fp8")
    got = dict(weights.hf_weights_iterator(d, "cpu", dequantize_fp8=True))
    assert not any(k.endswith("weight_scale_inv") for k in got)
    with safe_open(os.path.join(d, "model.safetensors"), framework="pt") as f:
        for key in ("model.layers.0.self_attn.q_proj", "model.layers.0.mlp.experts.1.down_proj"):
            w8, s = f.get_tensor(key + ".weight"), f.get_tensor(key + ".weight_scale_inv")
            want = (w8.float() * s.repeat_interleave(128, 0).repeat_interleave(128, 1)[: w8.shape[0], : w8.shape[1]]).half()
            assert got[key + ".weight"].dtype == torch.float16 and torch.equal(got[key + ".weight"], want)
    raw = dict(weights.hf_weights_iterator(d, "cpu", dequantize_fp8=False))
    assert raw["model.layers.0.self_attn.q_proj.weight"].dtype == torch.uint8
    assert "model.layers.0.self_attn.q_proj.weight_scale_inv" in raw


def test_config_json_parsing():
    with open(os.path.join(GOLDEN, "ckpt", "qwen3_moe_fp8", "config.json")) as f:
        cfg = json.load(f)
    geo = weights.geometry_from_hf_config(cfg)
    assert (geo.num_experts, geo.moe_intermediate_size, geo.head_dim, geo.use_qk_norm, geo.qkv_bias) == (2, 256, 32, True, False)
    quant, int4 = weights.quantization_from_hf_config(cfg)
    assert quant.format == "fp8" and (quant.group_n, quant.group_k) == (128, 128) and int4 is None
    awq = {"quantization_config": {"quant_method": "awq", "bits": 4, "group_size": 64, "version": "GEMM", "zero_point": True}}
    assert weights.quantization_from_hf_config(awq) == (None, weights.Int4Checkpoint("awq", 64))
    gptq = {"quantization_config": {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": True}}
    assert weights.quantization_from_hf_config(gptq) == (None, weights.Int4Checkpoint("gptq", 128))
    gptq["quantization_config"]["checkpoint_format"] = "gptq_v2"
    assert weights.quantization_from_hf_config(gptq)[1].fmt == "gptq_v2"
    with pytest.raises(ValueError, match="only 4-bit"):
        weights.quantization_from_hf_config({"quantization_config": {"quant_method": "gptq", "bits": 8}})
    with pytest.raises(ValueError, match="unsupported quant_method"):
        weights.quantization_from_hf_config({"quantization_config": {"quant_method": "bitsandbytes"}})


def test_int4_checkpoint_cuts_follow_the_layer_kind(tp_state):
    """Column-parallel layers keep a block of the LAST dimension of every AWQ / GPTQ tensor, row-parallel ones a block
    of the FIRST; g_idx follows the input channels and is rebased to the rank's first group."""
    ps._TP_WORLD_SIZE, ps._TP_RANK = 2, 1
    k, n, g = 256, 64, 64
    qw_awq, qz, sc = torch.arange(k * n // 8).view(k, n // 8), torch.arange(k // g * n // 8).view(k // g, n // 8), \
        torch.arange(k // g * n).view(k // g, n)
    col, row = "layers.0.mlp.up_proj", "layers.0.mlp.down_proj"
    assert torch.equal(weights.tp_shard_int4(col + ".qweight", qw_awq, g), qw_awq[:, n // 16:])
    assert torch.equal(weights.tp_shard_int4(col + ".qzeros", qz, g), qz[:, n // 16:])
    assert torch.equal(weights.tp_shard_int4(col + ".scales", sc, g), sc[:, n // 2:])
    assert torch.equal(weights.tp_shard_int4(row + ".qweight", qw_awq, g), qw_awq[k // 2:])
    assert torch.equal(weights.tp_shard_int4(row + ".qzeros", qz, g), qz[k // g // 2:])
    gi = torch.arange(k) // g
    assert torch.equal(weights.tp_shard_int4(col + ".g_idx", gi, g), gi)
    assert torch.equal(weights.tp_shard_int4(row + ".g_idx", gi, g), torch.arange(k // 2) // g)
    assert torch.equal(weights.tp_shard_int4("norm_weight.qweight", qw_awq, g), qw_awq)  # replicated: untouched
    with pytest.raises(ValueError, match="does not divide"):
        weights.tp_shard_int4(row + ".qzeros", torch.zeros(3, 8), g)
