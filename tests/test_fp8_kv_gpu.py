"""fp8 (OCP e4m3) KV cache (extension -- the reference's pool is fp16): the quantising scatter against
torch's e4m3fn conversion bit for bit, attention over the fp8 pool against the pinned fp16 oracle on the widened pool at
the usual 1e-2, and the decode engine end to end (captured == eager; logits near the fp16-pool engine's)."""

import math

import pytest
import torch

import lite_llama_amd.kernels as K
from lite_llama_amd.executor import DecodeEngine
from lite_llama_amd.model import CausalLM, tiny_geometry
from lite_llama_amd.quantization import QuantConfig
from oracle import fp8_kv as F8
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("scales", [(1.0, 1.0), (0.5, 2.0), (0.37, 1.7)])
def test_quantising_scatter_equals_torch_e4m3fn(dtype, scales):
    g = torch.Generator().manual_seed(5)
    hk, hd, tokens, rows = 2, 64, 37, 80
    vals = (torch.randn(tokens, 2 * hk, hd, generator=g) * 3).to(dtype)
    # the corners: zeros, subnormal range (< 2^-6), exact ties between codes, the clamp
    vals[0, 0, :8] = torch.tensor([0.0, -0.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0009765625 * 3, 1e-5, -1e-5]).to(dtype)
    vals[1, 1, :8] = torch.tensor([1.0625, 1.1875, 17.0, 18.0, 19.0, 21.0, 27.0, 29.0]).to(dtype)   # halfway points
    vals[2, 2, :6] = torch.tensor([448.0, 449.0, 600.0, -1000.0, 30000.0, -465.0]).to(dtype)
    sel = torch.randperm(rows, generator=g)[:tokens].int()
    want = torch.zeros(rows, 2 * hk, hd, dtype=torch.uint8)
    want[sel.long()] = F8.quantize_rows(vals, hk, *scales)
    for pool_dtype in (torch.uint8, torch.float8_e4m3fn):
        pool = torch.zeros(rows, 2 * hk, hd, dtype=torch.uint8, device=DEV).view(pool_dtype)
        K.update_kv_buffer_fp8(vals.to(DEV), sel.to(DEV), pool, hk, *scales)
        assert torch.equal(pool.view(torch.uint8).cpu(), want)


@pytest.mark.parametrize("d,hq,hkv,lens,scales", [
    (128, 28, 4, [1, 128, 129, 600], (1.0, 1.0)),          # grouped workgroups, GQA 7
    (128, 14, 2, [300, 1500, 77], (0.37, 1.7)),            # > 1024 tokens: the counter-merged form
    (64, 8, 2, [40, 256, 513], (0.5, 2.0)),
    (128, 4, 4, [5], (1.0, 1.0)),                          # single partition
])
def test_attention_over_the_fp8_pool_equals_the_oracle_on_the_widened_pool(d, hq, hkv, lens, scales):
    g = torch.Generator().manual_seed(11)
    bsz, lmax = len(lens), max(lens)
    rows = bsz * lmax + 7
    kvf = torch.randn(rows, 2 * hkv, d, generator=g).half() * 1.5
    codes = F8.quantize_rows(kvf, hkv, *scales)
    k32, v32 = F8.widen(codes[:, :hkv], scales[0]), F8.widen(codes[:, hkv:], scales[1])
    q = (torch.randn(bsz, hq, d, generator=g) * 0.5).half()
    table = torch.zeros(bsz, lmax, dtype=torch.int32)
    perm = torch.randperm(rows, generator=g).int()
    for i, n in enumerate(lens):
        table[i, :n] = perm[i * lmax : i * lmax + n]
    req = torch.arange(bsz, dtype=torch.int32)
    seq = torch.tensor(lens, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    ref = O.flash_decoding(q.float(), k32, v32, scale, table, req, seq, lmax)
    pool = codes.to(DEV)
    out = K.flash_decoding_fp8kv(q.to(DEV), pool[:, :hkv], pool[:, hkv:], scale, table.to(DEV), req.to(DEV), seq.to(DEV),
                                 lmax, *scales)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=1e-2, atol=1e-2)
    # and the write path feeds it: scatter the fp16 rows with the kernel, same answer
    pool2 = torch.zeros_like(pool)
    K.update_kv_buffer_fp8(kvf.to(DEV), torch.arange(rows, dtype=torch.int32, device=DEV), pool2, hkv, *scales)
    assert torch.equal(pool2, pool)


def test_rejects_what_it_does_not_serve():
    pool = torch.zeros(64, 4, 32, dtype=torch.uint8, device=DEV)
    q = torch.zeros(1, 4, 32, dtype=torch.float16, device=DEV)
    table = torch.zeros(1, 8, dtype=torch.int32, device=DEV)
    one = torch.ones(1, dtype=torch.int32, device=DEV)
    with pytest.raises(Exception, match="LL_ERR_SHAPE"):   # head size 32
        K.flash_decoding_fp8kv(q, pool[:, :2], pool[:, 2:], 0.1, table, one - 1, one, 8)
    with pytest.raises(AssertionError):                    # bf16 queries
        K.flash_decoding_fp8kv(q.bfloat16(), pool[:, :2], pool[:, 2:], 0.1, table, one - 1, one, 8)
    with pytest.raises(Exception, match="LL_ERR_ARG"):     # a scale must be positive
        K.update_kv_buffer_fp8(q.expand(1, 4, 32).contiguous(), one - 1, pool, 2, 0.0, 1.0)


def test_decode_engine_with_an_fp8_pool():
    geo = tiny_geometry(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                        vocab_size=512, qkv_bias=True)
    quant = QuantConfig.int4_groupwise(128)
    model = CausalLM(geo, quant).init_synthetic(seed=21, quant=quant, device=DEV)
    ids = torch.randint(0, 512, (3, 21), generator=torch.Generator().manual_seed(8)).to(DEV)
    lens = torch.tensor([21, 9, 16], device=DEV)

    def run(kv_dtype, use_graph):
        eng = DecodeEngine(model, max_batch=3, max_seq_len=64, kv_dtype=kv_dtype)
        grabbed = []
        orig = model.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            grabbed.append(out.detach().float().cpu())
            return out

        first = eng.prefill(ids, lens)
        if not use_graph:                     # (a host copy cannot happen inside a capture)
            model.forward = spy
        try:
            toks = eng.decode(first, 12, use_graph=use_graph).cpu()
        finally:
            model.forward = orig
        return first.cpu(), toks, grabbed[0] if grabbed else None

    f16_first, f16_toks, f16_logits = run(torch.float16, False)
    e_first, e_toks, e_logits = run(torch.float8_e4m3fn, False)
    g_first, g_toks, _ = run(torch.float8_e4m3fn, True)
    assert torch.equal(e_first, f16_first)            # the prefill attention reads the fresh fp16 q / k / v
    assert torch.equal(g_toks, e_toks)                # captured == eager on the fp8 pool
    # e4m3 keeps 3 mantissa bits of every cached K / V value: the logits move, by little
    err = (e_logits - f16_logits).abs().max().item()
    assert err <= 0.06 * f16_logits.abs().max().item() + 0.02, err
