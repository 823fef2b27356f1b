"""Pin the CPU oracle against golden vectors produced by the reference's own
Triton kernels (tests/golden/gen_golden.py, run under TRITON_INTERPRET=1).

Bit-exact where the contract is integer/byte/index work; the reference's own test
tolerances for floating-point reductions.
"""

import pytest
import torch

from oracle import oracle as O
from tests import _golden as G


def close(a, b, tol):
    torch.testing.assert_close(a.float(), b.float(), rtol=tol, atol=tol)


@pytest.mark.parametrize("name", G.names("skip_rmsnorm_"))
def test_skip_rmsnorm(name):
    d = G.load(name)
    r = d["r_in"].clone() if d["has_res"] else None
    y, r_out = O.skip_rmsnorm(d["x"].clone(), r, d["w"], d["eps"])
    # same arithmetic as the interpreter -> expect (near) bit equality
    if d["y_valid"]:
        close(y, d["y"], 2e-3)
    if d["has_res"]:
        if r_out.dtype == torch.bfloat16:
            # the interpreter truncates fp32->bf16 (no RNE); allow one bf16 ulp
            close(r_out, d["r_out"], 1e-2)
        else:
            assert torch.equal(r_out, d["r_out"])
        assert r_out.data_ptr() == r.data_ptr()


@pytest.mark.parametrize("name", G.names("swiglu_"))
def test_swiglu(name):
    d = G.load(name)
    close(O.swiglu_forward(d["a"], d["b"]), d["c"], 2e-3)


@pytest.mark.parametrize("name", G.names("rope_"))
def test_rope(name):
    d = G.load(name)
    q, k = O.rope_emb_forward(d["q"].clone(), d["k"].clone(), d["cos"], d["sin"], d["bs"], d["sl"])
    close(q, d["q_out"], 2e-3)
    close(k, d["k_out"], 2e-3)


def test_update_kv_buffer():
    d = G.load("update_kv_buffer")
    buf = d["buf_in"].clone()
    O.update_kv_buffer(d["vals"], d["idx"], buf)
    assert torch.equal(buf, d["buf_out"])


def test_update_kv_index():
    d = G.load("update_kv_index")
    t = d["table_in"].clone()
    O.update_kv_index(t, d["req"], d["seq"], d["sel"])
    assert torch.equal(t, d["table_out"])


@pytest.mark.parametrize("name", G.names("flash_decoding_"))
def test_flash_decoding(name):
    d = G.load(name)
    out = O.flash_decoding(d["q"], d["k_cache"], d["v_cache"], d["scale"], d["table"],
                           d["req_idx"], d["seq_len"], d["max_len"])
    close(out, d["out"], 1e-2 if out.dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("name", G.names("fa2_nopad_"))
def test_fa2_nopad(name):
    d = G.load(name)
    out = O.flash_attention2_no_pad(d["q"], d["k"], d["v"], d["sm_scale"], d["b_start_loc"],
                                    d["b_seq_len"], d["max_seq_len"])
    close(out, d["out"], 2e-3)


@pytest.mark.parametrize("name", G.names("w4a16_"))
def test_w4a16(name):
    d = G.load(name)
    # bit-exact parts: nibble unpack through the quantiser round trip
    qw, sc, zr = O.quantize_int4_groupwise(d["w_fp32"], d["group_size"])
    assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"]) and torch.equal(zr, d["zeros"])
    y = O.w4a16_matmul(d["x"], d["qweight"], d["scales"], d["zeros"], group_size=d["group_size"],
                       bias=d.get("bias"))
    close(y, d["y"], 2e-3)


@pytest.mark.parametrize("name", G.names("w8a16_"))
def test_w8a16(name):
    d = G.load(name)
    y = O.w8a16_matmul(d["x"], d["qweight"], d["scales"], group_n=d["group_n"], group_k=d["group_k"],
                       bias=d.get("bias"))
    close(y, d["y"], 2e-3)
    if name == "w8a16_int8_chan":
        qw, sc = O.quantize_int8_per_channel(d["w_fp32"])
        assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"])
    if name == "w8a16_int8_group":
        qw, sc = O.quantize_int8_groupwise(d["w_fp32"], 128)
        assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"])


def test_quantize_fp8():
    d = G.load("quantize_fp8_per_channel")
    qw, sc = O.quantize_fp8_per_channel(d["w_fp32"])
    assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"])


def test_smoothquant():
    d = G.load("smoothquant")
    acc, qa, a_scale = O.smoothquant_int32_acc(d["x"], d["qweight"])
    assert torch.equal(qa, d["qa"])  # truncating quantiser, bit-exact
    assert torch.equal(a_scale, d["a_scale"])
    y = O.smoothquant_matmul(d["x"], d["qweight"], d["scales"], bias=d["bias"])
    close(y, d["y"], 2e-3)


@pytest.mark.parametrize("name", G.names("moe_align_"))
def test_moe_align(name):
    d = G.load(name)
    s, e, n = O.moe_align_block_size(d["topk_ids"], d["block_size"], d["num_experts"])
    assert torch.equal(s, d["sorted_ids"])
    npost = int(d["num_post"][0])
    assert int(n[0]) == npost
    nblk = npost // d["block_size"]
    assert torch.equal(e[:nblk], d["expert_ids"][:nblk])
    assert torch.equal(e, d["expert_ids"])


@pytest.mark.parametrize("name", G.names("fused_moe_"))
def test_fused_moe(name):
    d = G.load(name)
    kw = {k: d[k] for k in ("w1_scale", "w2_scale", "group_n", "group_k") if k in d}
    out = O.fused_moe(d["x"], d["w1"], d["w2"], d["topk_weights"], d["topk_ids"], **kw)
    close(out, d["out"], 4e-3)


def test_quantisers():
    d = G.load("quantize_int4_groupwise")
    qw, sc, zr = O.quantize_int4_groupwise(d["w"], d["group_size"])
    assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"]) and torch.equal(zr, d["zeros"])
    # unpack is the bit-exact inverse of the packing
    nib = O.unpack_int4(qw)
    assert nib.min() >= 0 and nib.max() <= 15
    d = G.load("quantize_int8_per_channel")
    qw, sc = O.quantize_int8_per_channel(d["w"])
    assert torch.equal(qw, d["qweight"]) and torch.equal(sc, d["scales"])


# ------------------------------------------------------------------------------------- #
# sampler row: oracle vs vectors produced by the reference sampler
# ------------------------------------------------------------------------------------- #
def _sampler_golden():
    import numpy as np
    return np.load(G.GOLDEN_DIR + "/sampler_reference.npz")


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_oracle_repetition_penalty_matches_reference(dt):
    d = _sampler_golden()
    dtype = torch.float32 if dt == "f32" else torch.float16
    logits = torch.from_numpy(d[f"rp_{dt}.logits"]).to(dtype)
    ids, mask = torch.from_numpy(d[f"rp_{dt}.ids"]), torch.from_numpy(d[f"rp_{dt}.mask"])
    got = O.apply_repetition_penalty(logits, ids, mask, 1.3)
    assert torch.equal(got.float(), torch.from_numpy(d[f"rp_{dt}.scalar_1p3"]))
    got = O.apply_repetition_penalty(logits, ids, mask, torch.from_numpy(d[f"rp_{dt}.row_penalty"]))
    assert torch.equal(got.float(), torch.from_numpy(d[f"rp_{dt}.per_row"]))


def test_oracle_top_p_distribution_matches_reference():
    d = _sampler_golden()
    dist = O.top_p_distribution(torch.from_numpy(d["topp.logits"]), torch.from_numpy(d["topp.temperature"]),
                                torch.from_numpy(d["topp.top_p"]))
    ref = torch.from_numpy(d["topp.dist"])
    assert torch.equal(dist > 0, ref > 0)  # same nucleus
    torch.testing.assert_close(dist, ref, rtol=1e-5, atol=1e-7)
    # inverse CDF in token order really samples from it
    u = torch.tensor([0.0, 0.1, 0.5, 0.77, 0.999, 0.3])
    tok = O.sample_from_distribution(ref, u)
    assert bool((ref[torch.arange(6), tok] > 0).all())
