"""GPU parity tests of the third-generation W4A16 decode engine (csrc/gemm_w4_v3.hip): weights streamed
from their load-time layout (``pack_w4a16_weights``).  Same contract as ``w4a16_matmul``
(lite_llama/kernels/quantization/w4a16.py:152-207): nibble unpack bit-exact, results within the
reference's tolerance (5e-2) -- checked at 1e-2 -- of the CPU oracle, on the real Qwen2.5-7B shapes
(every split the host plan can produce) and on a randomised shape sweep against the reference-format
engine."""

import random

import pytest
import torch

from oracle import oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda"


def Q():
    import lite_llama_amd.kernels.quantization as q

    return q


def close(a, b, tol):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), rtol=tol, atol=tol)


def expected_pack(qw: torch.Tensor) -> torch.Tensor:
    """CPU restatement of the packer: block (tile, chunk) = [wave 8][lane 64][4 words]; wave (ng = w & 3,
    kh = w >> 2), lane (nl = l & 31, h = l >> 5) holds source words chunk*16 + kh*8 + h*4 + 0..3 of row
    tile*128 + ng*32 + nl; inside a word the even nibbles move to positions 0..3, the odd ones to 4..7."""
    n, kp = qw.shape
    nib = O.unpack_int4(qw).to(torch.int64)  # [N, K] natural order
    k = kp * 8
    oct_ = nib.reshape(n, k // 8, 8)
    order = [0, 2, 4, 6, 1, 3, 5, 7]
    words = sum(oct_[:, :, order[i]] << (4 * i) for i in range(8))  # [N, K/8] permuted words (as int64)
    words = words.reshape(n // 128, 4, 32, k // 128, 2, 2, 4)  # tile, ng, nl, chunk, kh, h, j
    words = words.permute(0, 3, 4, 1, 5, 2, 6)  # tile, chunk, kh, ng, h, nl, j
    words = words.reshape(n // 128, k // 128, 8, 64, 4)
    return (words & 0xFFFFFFFF).to(torch.int64)


def test_pack_weights_bit_exact():
    g = torch.Generator().manual_seed(5)
    qw = torch.randint(-(2**31), 2**31 - 1, (384, 64), dtype=torch.int64, generator=g).to(torch.int32)
    got = Q().pack_w4a16_weights(qw.to(DEV)).cpu().to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, expected_pack(qw))
    with pytest.raises(ValueError):
        Q().pack_w4a16_weights(qw[:100].to(DEV))


def test_prepacked_unpack_bit_exact():
    """x = one-hot rows, scale 1, zero 0: the GEMM output IS the nibble matrix (every k position)."""
    n, k = 256, 256
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64).to(torch.int32)
    sc, zr = torch.ones(n, k // 128), torch.zeros(n, k // 128)
    pw = Q().pack_w4a16_weights(qw.to(DEV))
    ps = Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    nib = O.unpack_int4(qw).float()
    for base in range(0, k, 64):
        x = torch.eye(k, dtype=torch.float16)[base:base + 64]
        y = Q().w4a16_matmul_prepacked(x.to(DEV), pw, ps, group_size=128)
        assert torch.equal(y.float().cpu(), nib[:, base:base + 64].T)


@pytest.mark.parametrize("name", G.names("w4a16_"))
def test_prepacked_golden(name):
    d = G.load(name)
    n, kp = d["qweight"].shape
    gs = int(d["group_size"])
    if n % 128 or (kp * 8) % 128 or gs % 128 or d["x"].reshape(-1, kp * 8).shape[0] > 64:
        pytest.skip("fixture shape is outside the decode engine (served by the reference-format engine)")
    pw = Q().pack_w4a16_weights(d["qweight"].to(DEV))
    ps = Q().pack_w4a16_scales(d["scales"].to(DEV), d["zeros"].to(DEV))
    bias = d.get("bias")
    y = Q().w4a16_matmul_prepacked(d["x"].to(DEV), pw, ps, group_size=gs, bias=None if bias is None else bias.to(DEV))
    close(y, d["y"], 1e-2)


@pytest.mark.parametrize("tb", [1, 2])
@pytest.mark.parametrize("M,N,K_,gs", [(64, 37888, 3584, 128), (64, 3584, 18944, 128), (64, 4608, 3584, 128),
                                       (64, 3584, 3584, 128), (17, 1024, 3584, 128), (64, 4608, 3584, 256),
                                       (1, 128, 512, 128), (32, 18944, 3584, 128), (64, 128, 128, 128)])
def test_prepacked_qwen_shapes_vs_oracle(M, N, K_, gs, tb):
    if tb == 2 and N % 256:
        pytest.skip("256-row tiles need N % 256 == 0")
    g = torch.Generator().manual_seed(N + K_ + M)
    x = torch.randn(M, K_, dtype=torch.float16, generator=g) * 0.5
    qw = torch.randint(-(2**31), 2**31 - 1, (N, K_ // 8), dtype=torch.int64, generator=g).to(torch.int32)
    sc = torch.rand(N, K_ // gs, generator=g) * 0.01 + 0.005
    zr = torch.randint(0, 16, (N, K_ // gs), generator=g).float()
    bias = (torch.randn(N, generator=g) * 0.1).half()
    pw = Q().pack_w4a16_weights(qw.to(DEV))
    ps = Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    xd = x.to(DEV)
    y0 = Q().w4a16_matmul_prepacked(xd, pw, ps, group_size=gs, bias=bias.to(DEV), _tile_blocks=tb)
    for _ in range(3):  # repeated launches reuse the split-K scratch (counters must come back to zero)
        assert torch.equal(y0, Q().w4a16_matmul_prepacked(xd, pw, ps, group_size=gs, bias=bias.to(DEV), _tile_blocks=tb))
    rows = torch.randperm(N, generator=g)[:256].sort().values
    ref = O.w4a16_matmul(x, qw[rows], sc[rows], zr[rows], group_size=gs, bias=bias[rows])
    close(y0[:, rows.to(DEV)], ref, 1e-2)


def test_prepacked_float_zero_points_at_1e2():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 512, dtype=torch.float16, generator=g)
    qw = torch.randint(-(2**31), 2**31 - 1, (256, 64), dtype=torch.int64, generator=g).to(torch.int32)
    sc = torch.rand(256, 4, generator=g) * 0.02 + 0.01
    zr = torch.rand(256, 4, generator=g) * 15  # the format stores zeros as floats: non-integers must work
    ref = O.w4a16_matmul(x, qw, sc, zr, group_size=128)
    y = Q().w4a16_matmul_prepacked(x.to(DEV), Q().pack_w4a16_weights(qw.to(DEV)),
                                   Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV)), group_size=128)
    close(y, ref, 1e-2)
    y2 = Q().w4a16_matmul(x.to(DEV), qw.to(DEV), sc.to(DEV), zr.to(DEV), group_size=128)
    close(y2, ref, 1e-2)


def test_prepacked_gate_up_swiglu_equals_two_step():
    """The fused epilogue on rows interleaved gate/up == swiglu(matmul(gate), matmul(up)) of the same engine."""
    import lite_llama_amd.kernels as K
    g = torch.Generator().manual_seed(9)
    i_sz, k = 1024, 1536
    x = (torch.randn(64, k, generator=g) * 0.5).half().to(DEV)
    qw = torch.randint(-(2**31), 2**31 - 1, (2 * i_sz, k // 8), dtype=torch.int64, generator=g).to(torch.int32).to(DEV)
    sc = (torch.rand(2 * i_sz, k // 128, generator=g) * 0.01 + 0.005).to(DEV)
    zr = torch.randint(0, 16, (2 * i_sz, k // 128), generator=g).float().to(DEV)
    fused = Q().w4a16_matmul_prepacked(x, Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr), gate_up_swiglu=True)
    fused2 = Q().w4a16_matmul_prepacked(x, Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr), gate_up_swiglu=True,
                                        _tile_blocks=2)
    gate = Q().w4a16_matmul_prepacked(x, Q().pack_w4a16_weights(qw[0::2].contiguous()),
                                      Q().pack_w4a16_scales(sc[0::2].contiguous(), zr[0::2].contiguous()))
    up = Q().w4a16_matmul_prepacked(x, Q().pack_w4a16_weights(qw[1::2].contiguous()),
                                    Q().pack_w4a16_scales(sc[1::2].contiguous(), zr[1::2].contiguous()))
    two = K.swiglu_forward(gate, up)
    close(fused, two, 2e-3)  # same rounding points; only the fp32 summation order of the split can differ
    close(fused2, two, 2e-3)


def test_prepacked_random_shapes_match_reference_format_engine():
    """Plan edge cases (stream-K vs tile groups, 1..12 contributors per tile, M = 1..64, K/N from one unit to
    hundreds): equal to the reference-format engines up to the fp32 summation order; repeated launches
    bit-identical."""
    import lite_llama_amd.kernels as K
    rnd = random.Random(13)
    for it in range(40):
        m = rnd.choice([1, 2, 7, 16, 31, 32, 33, 48, 63, 64])
        n = 128 * rnd.choice([1, 2, 3, 5, 8, 9, 17, 28, 36, 61, 148, 255, 256, 257, 300])
        k = 128 * rnd.choice([1, 2, 3, 4, 5, 7, 8, 13, 28, 29, 37, 64, 148])
        gs = rnd.choice([128, 128, 256]) if k % 256 == 0 else 128
        g = torch.Generator(device=DEV).manual_seed(it)
        x = (torch.randn(m, k, generator=g, device=DEV) * 0.5).half()
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, generator=g, device=DEV).to(torch.int32)
        sc = torch.rand(n, k // gs, generator=g, device=DEV) * 0.01 + 0.005
        zr = torch.randint(0, 16, (n, k // gs), generator=g, device=DEV).float()
        bias = (torch.randn(n, generator=g, device=DEV) * 0.1).half() if it % 3 == 0 else None
        pw, ps = Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr)
        tb = 2 if (n % 256 == 0 and it % 2 == 0) else 1  # both tile widths of the engine
        y3 = Q().w4a16_matmul_prepacked(x, pw, ps, group_size=gs, bias=bias, _tile_blocks=tb)
        y3b = Q().w4a16_matmul_prepacked(x, pw, ps, group_size=gs, bias=bias, _tile_blocks=tb)
        y2 = K.w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias)
        scale = y2.float().abs().max().item() + 1e-6
        assert torch.equal(y3, y3b), (m, n, k, gs, tb)
        assert (y3.float() - y2.float()).abs().max().item() <= 2e-3 * scale + 2e-3, (m, n, k, gs, tb)


@pytest.mark.parametrize("M,N,K_", [(64, 3584, 3584), (64, 3584, 18944), (17, 1024, 3584), (64, 4608, 3584), (1, 128, 512)])
def test_partial_sums_and_reducing_norm(M, N, K_):
    """Split-K partial mode: sum of the partials == the ordinary epilogue's output (fp32 order aside), and
    skip_rmsnorm_partials == skip_rmsnorm of that output (same arithmetic from the rounded x on)."""
    import lite_llama_amd.kernels as K
    from lite_llama_amd.kernels.norm_act import skip_rmsnorm_partials
    g = torch.Generator().manual_seed(N + K_)
    x = (torch.randn(M, K_, generator=g) * 0.5).half().to(DEV)
    qw = torch.randint(-(2**31), 2**31 - 1, (N, K_ // 8), dtype=torch.int64, generator=g).to(torch.int32).to(DEV)
    sc = (torch.rand(N, K_ // 128, generator=g) * 0.01 + 0.005).to(DEV)
    zr = torch.randint(0, 16, (N, K_ // 128), generator=g).float().to(DEV)
    pw, ps = Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr)
    full = Q().w4a16_matmul_prepacked(x, pw, ps)
    parts = Q().w4a16_matmul_partials(x, pw, ps)
    assert parts is not None and parts.parts.shape[1:] == (M, N)
    scale = full.float().abs().max().item()
    assert (parts.materialise().float() - full.float()).abs().max().item() <= 2e-3 * scale + 2e-3
    res = (torch.randn(M, N, generator=g) * 0.5).half().to(DEV)
    wn = (1 + 0.1 * torch.randn(N, generator=g)).half().to(DEV)
    y_ref, r_ref = K.skip_rmsnorm(parts.materialise(), res.clone(), wn, 1e-6)
    r_in = res.clone()
    y, r = skip_rmsnorm_partials(parts, r_in, wn, 1e-6)
    assert r.data_ptr() == r_in.data_ptr()
    # x is rounded from the same fp32 sums up to the summation order: an occasional 1-ulp flip of x is allowed
    close(r, r_ref, 2e-3)
    close(y, y_ref, 4e-3)
    exact = (r == r_ref).float().mean().item()
    assert exact > 0.98, exact


def _short_plan(m, n, k, g=128):
    import ctypes
    from lite_llama_amd import _lib
    out = (ctypes.c_int32 * 8)()
    assert _lib.lib().ll_w4a16_short_plan(m, n, k, g, out) == 0
    return dict(zip(("takes", "grid", "R", "S", "P", "kb_base", "kb_rem", "lds"), out))


def test_short_stream_unpack_bit_exact_at_every_k_position():
    """Round 6, csrc/gemm_short.hip: x = one-hot rows, scale 1, zero 0 -> the sum of the engine's planes IS the nibble matrix, for
    every k position of every k-slice (route asserted: the short-stream engine takes the launch)."""
    n, k = 512, 1024
    assert _short_plan(64, n, k)["takes"] == 1
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64).to(torch.int32)
    sc, zr = torch.ones(n, k // 128), torch.zeros(n, k // 128)
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    nib = O.unpack_int4(qw).float()
    for k0 in range(0, k, 64):
        x = torch.zeros(64, k, dtype=torch.float16)
        x[torch.arange(64), k0 + torch.arange(64)] = 1.0
        parts = Q().w4a16_matmul_partials(x.to(DEV), pw, ps)
        assert parts.parts.shape[0] == _short_plan(64, n, k)["S"]
        got = parts.parts.sum(0).cpu()
        assert torch.equal(got, nib[:, k0:k0 + 64].t().contiguous()), k0


def _short_full_plan(m, n, k, gs=128):
    import ctypes
    import lite_llama_amd._lib as L_
    out = (ctypes.c_int32 * 8)()
    assert L_.lib().ll_w4a16_short_full_plan(m, n, k, gs, out) == 0
    return dict(zip(("takes", "items", "R", "MT", "halves", "P", "slots", "lds"), out))


def test_short_stream_full_k_unpack_bit_exact_at_every_k_position():
    """Round 6, csrc/gemm_short_full.hip (finished outputs, all of K per workgroup, activations through a ring): x = one-hot rows,
    scale 1, zero 0 -> the output IS the nibble matrix, for every k position -- K long enough that the ring wraps (36 chunks through
    16 / 18 slots) -- for both batch halves and for a batch of 16 rows (whole-batch items); route asserted."""
    n, k = 1024, 4608
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64).to(torch.int32)
    sc, zr = torch.ones(n, k // 128), torch.zeros(n, k // 128)
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    nib = O.unpack_int4(qw).float()
    for m in (64, 16):
        p = _short_full_plan(m, n, k)
        assert p["takes"] == 1 and p["halves"] == (2 if m > 32 else 1) and k // 128 > p["slots"], p
        for k0 in range(0, k, m):
            x = torch.zeros(m, k, dtype=torch.float16)
            x[torch.arange(m), k0 + torch.arange(m)] = 1.0
            got = Q().w4a16_matmul_prepacked(x.to(DEV), pw, ps).float().cpu()
            assert torch.equal(got, nib[:, k0:k0 + m].t().contiguous()), (m, k0)


@pytest.mark.parametrize("M", [1, 17, 32, 33, 64])
@pytest.mark.parametrize("N,K_,gs,swiglu", [(3584, 3584, 128, False), (1024, 3584, 128, False), (4608, 3584, 128, True), (9472, 3584, 128, True),
                                            (2048, 1536, 128, False), (4096, 4096, 256, False), (8192, 4096, 128, True), (512, 128, 128, False),
                                            (1024, 256, 128, True), (640, 384, 128, False)])
def test_short_stream_full_k_outputs_match_oracle_and_unit_loop(M, N, K_, gs, swiglu):
    """The all-of-K short-stream form at the shapes it serves -- the reference-shaped layer's q / o / k|v projections, the fused
    gate|up of TP 4 / 8 shards (swiglu epilogue on interleaved rows), other models' widths, every batch class incl. ragged halves --
    against the unit loop (forced; same dequantiser, only the fp32 summation order differs: outputs equal up to one fp16 ulp of the
    largest value) and the CPU oracle at 1e-2 (reference tolerance 5e-2, w4a16.py:152-207); bias on the plain form."""
    g = torch.Generator().manual_seed(N * 3 + K_ + M)
    x = (torch.randn(M, K_, generator=g) * 0.5).half()
    qw = torch.randint(-(2**31), 2**31 - 1, (N, K_ // 8), dtype=torch.int64, generator=g).to(torch.int32)
    sc = torch.rand(N, K_ // gs, generator=g) * 0.01 + 0.005
    zr = torch.randint(0, 16, (N, K_ // gs), generator=g).float()
    bias = None if swiglu else (torch.randn(N, generator=g) * 0.1).half()
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    p = _short_full_plan(M, N, K_, gs)
    assert p["takes"] == 1, p
    kw = dict(group_size=gs, gate_up_swiglu=True) if swiglu else dict(group_size=gs, bias=bias.to(DEV))
    a = Q().w4a16_matmul_prepacked(x.to(DEV), pw, ps, **kw)
    b = Q().w4a16_matmul_prepacked(x.to(DEV), pw, ps, _tile_blocks=1, **kw)   # the unit loop, forced
    again = Q().w4a16_matmul_prepacked(x.to(DEV), pw, ps, **kw)
    assert torch.equal(a, again) and a.shape == (M, N // 2 if swiglu else N) and bool(torch.isfinite(a).all())
    scale = b.float().abs().max().item()
    assert (a.float() - b.float()).abs().max().item() <= scale * 2.0 ** -9 + 1e-6
    if M in (17, 64) and N * K_ <= 4608 * 3584:
        full = O.w4a16_matmul(x, qw, sc, zr, group_size=gs).float()
        if swiglu:
            f16 = full.half().float()
            ref = torch.nn.functional.silu(f16[:, 0::2]) * f16[:, 1::2]
        else:
            ref = full + bias.float()
        close(a, ref.half(), 3e-2 if swiglu else 1e-2)  # (silu(g) * u multiplies two fp16-rounded sums: W4A16's own bar is 5e-2)


@pytest.mark.parametrize("M", [1, 17, 32, 33, 64])
@pytest.mark.parametrize("N,K_,gs", [(4608, 3584, 128), (3584, 3584, 128), (2304, 3584, 128), (3584, 2432, 128), (4736 // 128 * 128, 3584, 128),
                                     (5120, 2048, 128), (2048, 4096, 256), (1024, 512, 512), (128, 128, 128)])
def test_short_stream_planes_match_oracle_and_unit_loop(M, N, K_, gs):
    """The short-stream engine's planes (headline q|k|v / o, TP-shard and other-model shapes, ragged slices, every batch class)
    against the CPU oracle at 1e-2 (reference tolerance 5e-2, w4a16.py:152-207) and against the unit loop's planes (same
    dequantiser: only the fp32 summation order differs); rows >= M of the last batch half are never written."""
    g = torch.Generator().manual_seed(N * 7 + K_ + M)
    x = (torch.randn(M, K_, generator=g) * 0.5).half()
    qw = torch.randint(-(2**31), 2**31 - 1, (N, K_ // 8), dtype=torch.int64, generator=g).to(torch.int32)
    sc = torch.rand(N, K_ // gs, generator=g) * 0.01 + 0.005
    zr = torch.randint(0, 16, (N, K_ // gs), generator=g).float()
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    plan = _short_plan(M, N, K_, gs)
    a = Q().w4a16_matmul_partials(x.to(DEV), pw, ps, group_size=gs)
    b = Q().w4a16_matmul_partials(x.to(DEV), pw, ps, group_size=gs, _unit_loop_engine=True)
    assert a is not None and b is not None
    if plan["takes"]:
        assert a.parts.shape == (plan["S"], M, N)
    sa, sb = a.parts.sum(0), b.parts.sum(0)
    scale = sb.abs().max().item()
    assert (sa - sb).abs().max().item() <= 2e-5 * scale + 1e-6
    if N * K_ <= 4608 * 3584 and M in (17, 64):
        ref = O.w4a16_matmul(x, qw, sc, zr, group_size=gs).float()
        close(sa.half(), ref, 1e-2)


def test_short_stream_planes_feed_the_reducing_norm():
    """skip_rmsnorm_partials over the short-stream planes against skip_rmsnorm of their sum in plane order, rounded once (the
    projection's own epilogue rounding): the residual -- the value the next layer builds on -- bit for bit, the normalised row to
    the last ulp of the row statistic's summation order -- the consumer contract of epilogue 2, whatever the plane count."""
    import lite_llama_amd.kernels as K
    from lite_llama_amd.kernels.norm_act import skip_rmsnorm_partials
    M, N, K_ = 64, 3584, 3584
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K_, generator=g) * 0.5).half().to(DEV)
    qw = torch.randint(-(2**31), 2**31 - 1, (N, K_ // 8), dtype=torch.int64, generator=g).to(torch.int32).to(DEV)
    sc = (torch.rand(N, K_ // 128, generator=g) * 0.01 + 0.005).to(DEV)
    zr = torch.randint(0, 16, (N, K_ // 128), generator=g).float().to(DEV)
    pw, ps = Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr)
    parts = Q().w4a16_matmul_partials(x, pw, ps)
    assert parts.parts.shape[0] == _short_plan(M, N, K_)["S"]
    acc = torch.zeros(M, N, device=DEV)
    for s in range(parts.parts.shape[0]):  # the consumer's order: plane 0 first
        acc = acc + parts.parts[s]
    res = (torch.randn(M, N, generator=g) * 0.5).half().to(DEV)
    wn = (1 + 0.1 * torch.randn(N, generator=g)).half().to(DEV)
    y_ref, r_ref = K.skip_rmsnorm(acc.half(), res.clone(), wn, 1e-6)
    y, r = skip_rmsnorm_partials(parts, res.clone(), wn, 1e-6)
    assert torch.equal(r, r_ref)
    close(y, y_ref, 2e-3)
    assert (y == y_ref).float().mean().item() > 0.999


@pytest.mark.parametrize("dtype_bias", [True, False])
@pytest.mark.parametrize("ctx", [200, 300, 549, 1000])
def test_decode_attention_over_qkv_partials(ctx, dtype_bias):
    """The one-launch decode attention fed with the fp32 split-K partials of the fused q|k|v projection == the same
    attention fed with the finished projection (Qwen2.5-7B head geometry): pool rows of the new token and outputs agree
    up to the rounding of x (the sums are the same fp32 values up to the summation order)."""
    import lite_llama_amd.kernels as K
    from lite_llama_amd.kernels.attention import decode_attention, decode_attention_partials
    HQ, HKV, D, B, HID = 28, 4, 128, 16, 3584
    N = (HQ + 2 * HKV) * D
    g = torch.Generator().manual_seed(ctx)
    x = (torch.randn(B, HID, generator=g) * 0.5).half().to(DEV)
    qw = torch.randint(-(2**31), 2**31 - 1, (N, HID // 8), dtype=torch.int64, generator=g).to(torch.int32).to(DEV)
    sc = (torch.rand(N, HID // 128, generator=g) * 0.004 + 0.002).to(DEV)
    zr = torch.randint(0, 16, (N, HID // 128), generator=g).float().to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).half().to(DEV) if dtype_bias else None
    pw, ps = Q().pack_w4a16_weights(qw), Q().pack_w4a16_scales(sc, zr)
    parts = Q().w4a16_matmul_partials(x, pw, ps)
    assert parts is not None
    full = Q().w4a16_matmul_prepacked(x, pw, ps, bias=bias)
    rows = B * (ctx + 1)
    pool_a = (torch.randn(rows, 2 * HKV, D, generator=g) * 0.5).half().to(DEV)
    pool_b = pool_a.clone()
    table = torch.randperm(rows, generator=g).int().view(B, ctx + 1).to(DEV)
    sel = table[:, ctx].contiguous()
    seq = torch.full((B,), ctx + 1, dtype=torch.int32, device=DEV)
    req = torch.arange(B, dtype=torch.int32, device=DEV)
    pos = torch.full((B,), ctx, dtype=torch.int64, device=DEV)
    cos = torch.randn(ctx + 8, D, generator=g).half().to(DEV)
    sin = torch.randn(ctx + 8, D, generator=g).half().to(DEV)
    scale = 1.0 / D**0.5
    q = full[:, : HQ * D].view(B, HQ, D)
    kv = full[:, HQ * D:].view(B, 2 * HKV, D)
    want = decode_attention(q, kv, cos, sin, pos, sel, pool_a, scale, table, req, seq, ctx + 1)
    got = decode_attention_partials(parts, bias, HQ, HKV, D, cos, sin, pos, sel, pool_b, scale, table, req, seq, ctx + 1)
    if ctx == 200:  # two partitions = 128 lanes x 2 slots < the 288 slots of (7 + 2) x 128 values: declined, caller falls back
        assert got is None
        return
    assert want is not None and got is not None
    new_a, new_b = pool_a[sel.long()], pool_b[sel.long()]
    close(new_b, new_a, 4e-3)
    assert (new_a == new_b).float().mean().item() > 0.97
    untouched = torch.ones(rows, dtype=torch.bool, device=DEV)
    untouched[sel.long()] = False
    assert torch.equal(pool_a[untouched], pool_b[untouched])
    close(got, want, 4e-3)


def test_sticky_merge_error_word_is_seen_from_an_indexless_device_and_only_on_this_streams_sets():
    """``gemm_scratch_error(torch.device("cuda"))`` -- what ``DecodeEngine`` passes by default -- matched no
    scratch set (its index is None, the sets are keyed by the tensors' device index), so a poisoned merge could never raise.
    A planted non-zero counter word is now reported through both spellings of the device; a word planted in ANOTHER eager
    stream's set (legitimately non-zero in the middle of that stream's GEMM) is not."""
    from lite_llama_amd import _lib as L

    n, k = 256, 512
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=DEV).to(torch.int32)
    pw = Q().pack_w4a16_weights(qw)
    ps = Q().pack_w4a16_scales(torch.rand(n, k // 128, device=DEV) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=DEV).float())
    x = torch.randn(8, k, device=DEV, dtype=torch.float16)
    Q().w4a16_matmul_prepacked(x, pw, ps, group_size=128)
    for d in (torch.device("cuda"), torch.device("cuda", torch.cuda.current_device())):
        assert L.gemm_scratch_error(d) is False
    key, _ = L.scratch_keys(torch.device("cuda"))
    assert key[1] == torch.cuda.current_device() and key in L._gemm_ws
    word = L._gemm_ws[key][1][7:8]
    word.fill_(1)
    try:
        for d in (torch.device("cuda"), torch.device("cuda", torch.cuda.current_device())):
            assert L.gemm_scratch_error(d) is True
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            Q().w4a16_matmul_prepacked(x, pw, ps, group_size=128)  # creates the side stream's own set
            assert L.gemm_scratch_error(torch.device("cuda")) is False  # this stream's sets are clean
    finally:
        word.zero_()
    torch.cuda.synchronize()
    assert L.gemm_scratch_error(torch.device("cuda")) is False


def test_unpack_is_the_exact_inverse_of_pack():
    """``ll_w4a16_unpack_weights`` (round 4: the load-time layout as the ONLY resident copy of the weights) restores the
    reference-format words bit for bit -- every shape class of the packer, incl. a strided source."""
    g = torch.Generator().manual_seed(3)
    for n, k in [(128, 128), (256, 384), (4608, 3584), (3584, 18944)]:
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, generator=g).to(torch.int32).to(DEV)
        back = Q().unpack_w4a16_weights(Q().pack_w4a16_weights(qw))
        assert back.shape == qw.shape and torch.equal(back, qw)
    with pytest.raises(ValueError):
        Q().unpack_w4a16_weights(qw)   # not a packed tensor


def test_compacted_model_keeps_one_copy_of_the_int4_weights_and_the_same_numbers():
    """``CausalLM.compact_weights()``: the parameters alias the decode engine's layout, the
    reference-format tensors are released and rebuilt on demand.  A small int4 model: prefill logits (80 rows: the generic
    engine over a rebuilt reference tensor) and greedy decode tokens (eager and captured: the pre-packed engine over the
    aliased storage) are BIT-equal before and after; the resident weight bytes shrink by the int4 payload; ``expand_weights``
    restores every parameter bit for bit; graphs captured BEFORE the compaction keep replaying (the packed storage did not
    move)."""
    import types

    from lite_llama_amd.executor import DecodeEngine
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    geo = tiny_geometry(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                        vocab_size=1024, qkv_bias=True)
    torch.manual_seed(21)
    model = CausalLM(geo)
    sd = {k: (torch.randn_like(v.float()) * (0.05 if v.dim() > 1 else 0.02) + (1.0 if "norm" in k else 0.0)).to(v.dtype)
          for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    model.quantize_(QuantConfig.int4_groupwise(128))
    ids = torch.randint(0, 1024, (2, 40), device=DEV)
    lens = torch.tensor([40, 37], dtype=torch.int32, device=DEV)

    def run(use_graph):
        eng = DecodeEngine(model, max_batch=2, max_seq_len=128, device=DEV)
        first = eng.prefill(ids, lens)
        toks = eng.decode(first, 6, use_graph=use_graph)
        torch.cuda.synchronize()
        return first.clone(), toks.clone(), eng

    def prefill_logits():
        B, LP = 2, 40
        sel = torch.arange(B * LP, dtype=torch.int32, device=DEV)
        table = torch.zeros(B, 64, dtype=torch.int32, device=DEV)
        for i, n in enumerate([40, 37]):
            table[i, :n] = sel[i * LP: i * LP + n]
        info = types.SimpleNamespace(kv_buffer=[torch.zeros(128, 4, 64, dtype=torch.float16, device=DEV) for _ in range(2)],
                                     cur_select_index=sel, b_req_tokens_table=table,
                                     b_start_loc=torch.arange(B, dtype=torch.int32, device=DEV) * LP,
                                     b_req_idx=torch.arange(B, dtype=torch.int32, device=DEV), b_seq_len=lens,
                                     max_actual_seq_len=LP)
        with torch.no_grad():
            out = model(ids, torch.arange(LP, device=DEV).unsqueeze(0).expand(B, LP), info)
        return torch.cat([out[0, :40], out[1, :37]]).clone()   # (pad positions are uninitialised, as in the reference)

    ref_params = {k: v.detach().clone() for k, v in model.named_parameters()}
    # compaction BEFORE any forward (what bench.py does: the merged q|k|v / gate|up storages do not exist yet) ...
    assert model.compact_weights() == sum(p.numel() * 4 for p in model.parameters() if p.dtype == torch.int32)
    logits_c = prefill_logits()
    first_c, toks_c, _ = run(True)
    model.expand_weights()
    for k, v in model.named_parameters():
        assert torch.equal(v, ref_params[k]), k
    # ... equals the uncompacted model
    logits0 = prefill_logits()
    assert torch.equal(logits_c, logits0)
    first0, toks0, eng_before = run(True)
    assert torch.equal(first_c, first0) and torch.equal(toks_c, toks0)
    bytes0 = model.weight_bytes()
    int4_bytes = sum(p.numel() * 4 for n, p in model.named_parameters() if p.dtype == torch.int32)
    freed = model.compact_weights()
    assert freed == int4_bytes and model.compact_weights() == 0          # every int4 parameter, once
    bytes1 = model.weight_bytes()
    assert bytes0 - bytes1 == int4_bytes, (bytes0, bytes1, int4_bytes)   # one copy of the payload is gone
    for k, v in model.named_parameters():
        assert v.shape == ref_params[k].shape and v.dtype == ref_params[k].dtype
    assert torch.equal(prefill_logits(), logits0)                        # > 64 rows: rebuilt reference tensor, generic engine
    for use_graph in (False, True):
        first1, toks1, _ = run(use_graph)
        assert torch.equal(first1, first0) and torch.equal(toks1, toks0)
    again = eng_before.decode(eng_before.prefill(ids, lens), 6, use_graph=True)   # the engine whose graph was captured BEFORE
    assert torch.equal(again, toks0)
    # state_dict() of the compacted model exports the reference layouts WITHOUT touching the model: same
    # storage, same epoch, a submodule's export alike; a submodule-level load into the aliased storage is refused
    epoch, ptrs = model.layout_epoch, {k: v.data_ptr() for k, v in model.named_parameters()}
    sd_c = model.state_dict()
    assert model.is_compacted() and model.layout_epoch == epoch
    assert all(v.data_ptr() == ptrs[k] for k, v in model.named_parameters())
    assert all(torch.equal(sd_c[k], ref_params[k]) for k in ref_params)
    sub = model.layers[0].self_attn.o_proj.state_dict()
    assert torch.equal(sub["weight"], ref_params["layers.0.self_attn.o_proj.weight"])
    with pytest.raises(RuntimeError, match="expand_weights"):
        model.layers[0].self_attn.o_proj.load_state_dict(sub)
    # replacing a parameter of a compacted model is refused loudly (its merged storage holds permuted words) ...
    layer0 = model.layers[0].self_attn
    keep = layer0.q_proj.weight
    layer0.q_proj.weight = torch.nn.Parameter(ref_params["layers.0.self_attn.q_proj.weight"].clone(), requires_grad=False)
    with pytest.raises(RuntimeError, match="expand_weights"):
        prefill_logits()
    layer0.q_proj._parameters["weight"] = keep
    layer0._qkv._key = layer0._qkv._snapshot()
    # ... after expand_weights() everything is the reference format again
    model.expand_weights()
    for k, v in model.named_parameters():
        assert torch.equal(v, ref_params[k]), k
    first2, toks2, _ = run(True)
    assert torch.equal(first2, first0) and torch.equal(toks2, toks0)


# ---------------------------------------------------------------------------------------------------------------- #
# the M-tiled engine for prefill-shaped calls (csrc/gemm_w4_prefill.hip, round 5)
# ---------------------------------------------------------------------------------------------------------------- #
def test_mtiled_unpack_bit_exact_at_every_k_position():
    """One-hot activation rows (more than 64 of them), scale 1, zero 0: the M-tiled GEMM's output IS the nibble matrix."""
    n, k = 512, 256
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64).to(torch.int32)
    sc, zr = torch.ones(n, k // 128), torch.zeros(n, k // 128)
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    x = torch.eye(k, dtype=torch.float16, device=DEV)  # 256 rows: row i picks k position i
    out = Q().w4a16_matmul_prepacked_rows(x, pw, ps)
    assert out is not None and torch.equal(out.cpu().float().T.contiguous(), O.unpack_int4(qw).float())


@pytest.mark.parametrize("m,n,k,gs", [(65, 256, 128, 128), (300, 512, 384, 128), (1000, 768, 512, 256), (257, 1024, 256, 128),
                                      (4096, 4608, 3584, 128)])
def test_mtiled_matches_oracle_and_generic_engine(m, n, k, gs):
    """Prefill-shaped calls (ragged row counts, both group sizes, bias, the fused gate|up epilogue) against the CPU oracle at
    1e-2 (the reference's own tolerance is 5e-2) and against the reference-layout engine; rows <= 64 of the same call equal
    the decode engine's rows up to the fp32 summation order."""
    g = torch.Generator().manual_seed(m + n + k)
    w = torch.randn(n, k, generator=g) * 0.05
    qw, sc, zr = O.quantize_int4_groupwise(w.half(), gs)
    x = (torch.randn(m, k, generator=g) * 0.5).half()
    bias = (torch.randn(n, generator=g) * 0.1).half()
    pw, ps = Q().pack_w4a16_weights(qw.to(DEV)), Q().pack_w4a16_scales(sc.to(DEV), zr.to(DEV))
    xd = x.to(DEV)
    got = Q().w4a16_matmul_prepacked_rows(xd, pw, ps, group_size=gs, bias=bias.to(DEV))
    assert got is not None and got.shape == (m, n)
    if m * n * k <= 2 ** 31:
        ref = O.w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias)
        close(got, ref, 1e-2)
    gen = Q().w4a16_matmul(xd, qw.to(DEV), sc.to(DEV), zr.to(DEV), group_size=gs, bias=bias.to(DEV))
    if m > 64:
        close(got, gen, 1e-2)
    dec = Q().w4a16_matmul_prepacked(xd[:64], pw, ps, group_size=gs, bias=bias.to(DEV))
    close(got[:64], dec, 2e-3)
    assert torch.equal(got, Q().w4a16_matmul_prepacked_rows(xd, pw, ps, group_size=gs, bias=bias.to(DEV)))  # deterministic
    # fused gate|up: rows interleaved (gate_j, up_j) -> silu(gate) * up, the arithmetic of swiglu_forward on the fp16 outputs
    sw = Q().w4a16_matmul_prepacked_rows(xd, pw, ps, group_size=gs, gate_up_swiglu=True)
    plain = Q().w4a16_matmul_prepacked_rows(xd, pw, ps, group_size=gs)
    from lite_llama_amd.kernels import swiglu_forward
    assert torch.equal(sw, swiglu_forward(plain[:, 0::2].contiguous(), plain[:, 1::2].contiguous()))
    # a strided activation view (row stride > k) and a shape off the 256-row grid
    wide = torch.zeros(m, k + 64, dtype=torch.float16, device=DEV)
    wide[:, :k] = xd
    assert torch.equal(Q().w4a16_matmul_prepacked_rows(wide[:, :k], pw, ps, group_size=gs, bias=bias.to(DEV)), got)
    if n > 256:
        assert Q().w4a16_matmul_prepacked_rows(xd, pw[: (n - 128) // 128], ps[:, : n - 128].contiguous(), group_size=gs) is None
