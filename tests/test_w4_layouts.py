"""AutoAWQ / AutoGPTQ int4 checkpoint tensors -> native W4A16 parameters.

Integer work, bit-exact.  CPU tier: the oracle's packers/converters against hand-computed words and
against the (reference-pinned) native dequantiser.  GPU tier: the HIP conversion == the oracle's,
and the converted layer through ``w4a16_matmul`` == the checkpoint's own dequantised matmul.
"""

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import w4_layouts as W


def _checkpoint(k, n, g, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 16, size=(k, n))
    z = rng.integers(1, 16, size=(k // g, n))        # 1..15: representable by GPTQ v1 (stores z - 1)
    s = (rng.random((k // g, n)) * 0.02 + 0.001).astype(np.float16)
    return q, z, s


def test_known_answer_words():
    # one AWQ word: channels 0..7 hold 0..7 -> nibbles (LSB first) 0,2,4,6,1,3,5,7
    q = np.arange(8).reshape(1, 8)
    qw, _ = W.awq_pack(q, q)
    assert qw.view(np.uint32)[0, 0] == 0x75316420
    # one GPTQ word: k = 0..7 hold 1..8 for a single column -> 0x87654321; zero 5 stored as 4 (v1)
    qk = (np.arange(8) + 1).reshape(8, 1).repeat(8, axis=1)
    zw = np.full((1, 8), 5)
    gw, gz = W.gptq_pack(qk, zw)
    assert gw.view(np.uint32)[0, 0] == 0x87654321 and gz.view(np.uint32)[0, 0] == 0x44444444
    assert W.gptq_pack(qk, zw, v1=False)[1].view(np.uint32)[0, 0] == 0x55555555
    # native word of row n: the same eight k values, sequential
    assert W.native_pack(qk.T.copy()).view(np.uint32)[0, 0] == 0x87654321


@pytest.mark.parametrize("fmt", ["awq", "gptq", "gptq_v2"])
def test_oracle_conversion_preserves_the_checkpoint_meaning(fmt):
    k, n, g = 256, 72, 64
    q, z, s = _checkpoint(k, n, g, 1)
    if fmt == "awq":
        nat = W.awq_to_native(*W.awq_pack(q, z), s, g)
    else:
        v1 = fmt == "gptq"
        nat = W.gptq_to_native(*W.gptq_pack(q, z, v1), s, g, v1)
    w_native = O.dequant_int4(torch.from_numpy(nat[0]), torch.from_numpy(nat[1]), torch.from_numpy(nat[2]), g)
    want = W.dequant_kn(q, z, s, g).T
    assert np.array_equal(w_native.numpy(), want)      # same fp32 arithmetic, so exactly equal


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["awq", "gptq", "gptq_v2"])
@pytest.mark.parametrize("shape", [(128, 8, 128), (256, 72, 64), (3584, 1024, 128), (1024, 3592, 32)])
def test_hip_conversion_is_bit_exact(fmt, shape):
    from lite_llama_amd.quantization import awq_to_w4a16, gptq_to_w4a16

    k, n, g = shape
    q, z, s = _checkpoint(k, n, g, 2)
    dev = "cuda"
    if fmt == "awq":
        qw, qz = W.awq_pack(q, z)
        want = W.awq_to_native(qw, qz, s, g)
        got = awq_to_w4a16(torch.from_numpy(qw).to(dev), torch.from_numpy(qz).to(dev), torch.from_numpy(s).to(dev), g)
    else:
        v1 = fmt == "gptq"
        qw, qz = W.gptq_pack(q, z, v1)
        want = W.gptq_to_native(qw, qz, s, g, v1)
        g_idx = torch.arange(k, dtype=torch.int32, device=dev) // g
        got = gptq_to_w4a16(torch.from_numpy(qw).to(dev), torch.from_numpy(qz).to(dev), torch.from_numpy(s).to(dev),
                            g_idx, g, fmt)
    for a, b, name in zip(got, want, ("qweight", "scales", "zeros")):
        assert a.dtype == torch.from_numpy(b).dtype
        assert np.array_equal(a.cpu().numpy(), b), name


@pytest.mark.gpu
def test_loaded_checkpoint_linear_runs_the_int4_path():
    from lite_llama_amd.linear import ReplicatedLinear
    from lite_llama_amd.quantization import load_int4_checkpoint_linear

    k, n, g = 512, 384, 128
    q, z, s = _checkpoint(k, n, g, 3)
    x = torch.randn(24, k, dtype=torch.float16, device="cuda")
    bias = (torch.randn(n) * 0.1).half().cuda()
    want = (x.float().cpu() @ torch.from_numpy(W.dequant_kn(q, z, s, g)) + bias.float().cpu()).half()
    for fmt in ("awq", "gptq"):
        qw, qz = W.awq_pack(q, z) if fmt == "awq" else W.gptq_pack(q, z)
        layer = ReplicatedLinear(k, n, bias=True).cuda()
        load_int4_checkpoint_linear(layer, torch.from_numpy(qw).cuda(), torch.from_numpy(qz).cuda(),
                                    torch.from_numpy(s).cuda(), fmt=fmt, group_size=g, bias=bias)
        assert layer.quant is not None and layer.weight.dtype == torch.int32
        torch.testing.assert_close(layer(x).cpu().float(), want.float(), rtol=5e-2, atol=5e-2)  # a8's tolerance


@pytest.mark.gpu
def test_conversion_rejects_what_it_cannot_represent():
    from lite_llama_amd.quantization import awq_to_w4a16, gptq_to_w4a16

    k, n, g = 256, 64, 128
    q, z, s = _checkpoint(k, n, g, 4)
    qw, qz = W.gptq_pack(q, z)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    with pytest.raises(NotImplementedError):
        gptq_to_w4a16(t(qw), t(qz), t(s), torch.randperm(k, device="cuda") // g, g)
    with pytest.raises(ValueError):
        gptq_to_w4a16(t(qw), t(qz), t(s).float(), None, g)
    with pytest.raises(ValueError):
        awq_to_w4a16(t(W.awq_pack(q, z)[0]), t(qz)[:1], t(s), g)
