"""CPU tier: the C-ABI library loads and exports every symbol include/lite_llama_amd.h declares
(no compute calls without a GPU), plus the host-side mirror of the reference's operator interface
(names, quant-method registry, parameter layout, error behaviour, engine metadata arithmetic)."""

import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_match_header():
    from lite_llama_amd import _lib, build

    build.build(verbose=False)
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "lite_llama_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t)\s+(ll_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ll_abi_version() == _lib.ABI_VERSION
    assert lib.ll_flash_decoding_num_partitions(129) == 2
    assert lib.ll_kv_alloc_scratch_bytes(4097) == 2 * (16 + 8) + 64


def test_w4a16_decode_plan_host_side():
    """Host-side launch plan of the pre-packed W4A16 engine (no device needed: without one the plan assumes 256 CUs): the
    split-K partial launches of the headline model -- q|k|v keeps five k-slices, o and down take the power-of-two split of
    the XCD-aware slice map (a launch that would fill < 85 % of the CUs keeps the plain split)."""
    from lite_llama_amd import _lib

    lib = _lib.lib()
    assert lib.ll_w4a16_prepacked_supported(64, 4608, 3584, 128) == 1
    assert lib.ll_w4a16_prepacked_supported(65, 4608, 3584, 128) == 0          # the decode engine stops at 64 rows
    assert lib.ll_w4a16_prepacked_supported(64, 4608, 3584, 96) == 0           # groups of 128 * 2^j
    # the unit loop's own split (epilogue 2 | 0x100 = the unit loop forced)
    counts = {(n, k): lib.ll_w4a16_partials_count_ex(64, n, k, 128, 2 | 0x100) for n, k in [(4608, 3584), (3584, 3584), (3584, 18944)]}
    assert counts == {(4608, 3584): 5, (3584, 3584): 8, (3584, 18944): 8}, counts
    assert all(1 <= c <= 12 for c in counts.values())                         # what ll_skip_rmsnorm_partials accepts
    assert lib.ll_w4a16_partials_count(64, 37888, 3584, 128) == 1             # many tiles: one plane (= no split)
    assert lib.ll_w4a16_partials_count(64, 3584, 18944, 128) == 8             # down: 149 KB per CU is not a short stream


def test_w4a16_short_stream_plan_host_side():
    """Host-side plan of the short-stream engine (csrc/gemm_short.hip, round 6; 256 CUs assumed without a device): it takes the
    split-K partial launches whose weight stream is a few tens of KB per CU -- the headline's q|k|v and o, TP shards -- with one
    round of workgroups, <= 8 planes (what ll_decode_attention_partials adds up), the k-slice of the activations resident in LDS;
    ll_w4a16_partials_count reports ITS plane count for those launches."""
    import ctypes
    from lite_llama_amd import _lib

    lib = _lib.lib()

    def plan(m, n, k, g=128):
        out = (ctypes.c_int32 * 8)()
        assert lib.ll_w4a16_short_plan(m, n, k, g, out) == 0
        return dict(zip(("takes", "grid", "R", "S", "P", "kb_base", "kb_rem", "lds"), out))

    for m in (1, 32, 33, 64):
        for n, k in [(4608, 3584), (3584, 3584), (2304, 3584), (3584, 1792), (3584, 2432), (5120, 2048), (2048, 4096)]:
            p = plan(m, n, k)
            assert p["takes"] == 1, (m, n, k, p)
            assert lib.ll_w4a16_partials_count(m, n, k, 128) == p["S"]
            assert 1 <= p["S"] <= 8 and p["R"] in (1, 2, 4, 8) and (n // 32) % p["R"] == 0
            items = (n // 32 // p["R"]) * p["S"]
            assert items <= 256 and p["grid"] == 8 * ((items + 7) // 8)       # one workgroup per CU, one round
            assert p["kb_base"] * p["S"] + p["kb_rem"] == k // 64            # the slices tile K in 64-k blocks
            blocks = p["kb_base"] + (1 if p["kb_rem"] else 0)
            assert (blocks + 8 // p["R"] - 1) // (8 // p["R"]) <= p["P"]     # pieces per consumer wave fit the template bound
            assert p["lds"] <= 160 * 1024 and p["lds"] >= (blocks // 2) * (2 if m > 32 else 1) * 32 * 256
    assert plan(64, 3584, 18944)["takes"] == 0 and plan(64, 37888, 3584)["takes"] == 0
    assert plan(65, 4608, 3584)["takes"] == 0 and plan(64, 4608, 3584, 96)["takes"] == 0


def test_w4a16_short_stream_full_k_plan_host_side():
    """Host-side plan of the short-stream engine's all-of-K form (csrc/gemm_short_full.hip, round 6; 256 CUs assumed without a
    device): narrow FINISHED outputs -- the reference-shaped layer's q / k|v / o calls, the fused gate|up of a TP >= 4 shard -- in one
    round of workgroups, 32-row batch halves when they fit, <= 16 weight pieces per consumer wave; wide outputs and long K are left to
    the row-group engine / unit loop."""
    import ctypes
    from lite_llama_amd import _lib

    lib = _lib.lib()

    def plan(m, n, k, g=128):
        out = (ctypes.c_int32 * 8)()
        assert lib.ll_w4a16_short_full_plan(m, n, k, g, out) == 0
        return dict(zip(("takes", "items", "R", "MT", "halves", "P", "slots", "lds"), out))

    for m, n, k, want in [(64, 3584, 3584, (1, 1, 2)), (64, 1024, 3584, (1, 1, 2)), (32, 3584, 3584, (1, 1, 1)), (64, 4608, 3584, (2, 1, 2)),
                          (64, 9472, 3584, (2, 2, 1)), (17, 4096, 4096, (1, 1, 1)), (64, 2048, 1536, (1, 1, 2)), (64, 8192, 4096, (2, 1, 2))]:
        p = plan(m, n, k)
        assert p["takes"] == 1 and (p["R"], p["MT"], p["halves"]) == want, (m, n, k, p)
        assert p["items"] == n // 32 // p["R"] * p["halves"] <= 256
        kq = 8 // p["R"]
        assert (k // 64 + kq - 1) // kq <= p["P"] <= 16 and p["P"] % 4 == 0
        assert p["slots"] >= kq and p["slots"] * p["MT"] * 32 * 256 <= p["lds"] <= 160 * 1024   # >= two rounds of chunk tiles
    for m, n, k in [(64, 37888, 3584), (64, 18944, 3584), (64, 3584, 18944), (65, 3584, 3584), (64, 256, 3584), (64, 3584, 9216)]:
        assert plan(m, n, k)["takes"] == 0, (m, n, k)


def test_kernel_names_match_reference_surface():
    import lite_llama_amd.kernels as k

    assert sorted(k.__all__) == sorted([
        "flash_attention2_no_pad", "flash_decoding", "fused_moe", "gelu", "leaky_relu", "moe_align_block_size",
        "relu", "rope_emb_forward", "skip_rmsnorm", "swiglu_forward", "tanh", "update_kv_buffer",
        "update_kv_index", "w4a16_matmul", "w8a16_matmul", "smoothquant_matmul"])


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors instead of routing anywhere else."""
    import lite_llama_amd.kernels as k

    x = torch.randn(2, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="GPU only"):
        k.skip_rmsnorm(x, None, torch.ones(64, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="GPU only"):
        k.swiglu_forward(x, x)


def test_validation_errors_precede_launch():
    import lite_llama_amd.kernels as k

    x = torch.randn(2, 256, dtype=torch.float16)
    qw = torch.zeros(8, 32, dtype=torch.int32)
    with pytest.raises(ValueError, match="must be fp16"):
        k.w4a16_matmul(x.float(), qw, torch.ones(8, 2), torch.zeros(8, 2))
    with pytest.raises(ValueError, match="int32"):
        k.w4a16_matmul(x, qw.long(), torch.ones(8, 2), torch.zeros(8, 2))
    with pytest.raises(ValueError, match="multiple of group_size"):
        k.w4a16_matmul(x, qw, torch.ones(8, 2), torch.zeros(8, 2), group_size=96)
    with pytest.raises(ValueError, match="uint8"):
        k.w8a16_matmul(x, torch.zeros(8, 256, dtype=torch.int16), torch.ones(8, 1), group_n=1, group_k=256)
    with pytest.raises(ValueError, match="group_k"):
        k.w8a16_matmul(x, torch.zeros(8, 256, dtype=torch.int8), torch.ones(8, 4), group_n=1, group_k=64)
    with pytest.raises(ValueError, match="int8"):
        k.smoothquant_matmul(x, torch.zeros(8, 256, dtype=torch.uint8), torch.ones(8))


def test_quant_registry_and_layout():
    from lite_llama_amd.quantization import (QuantConfig, SmoothQuantLinearMethod, W4A16LinearMethod,
                                             W8A16LinearMethod, W8A16MoeMethod, get_linear_method, get_moe_method)
    from lite_llama_amd.linear import LinearBase

    assert isinstance(get_linear_method(QuantConfig.int4_groupwise()), W4A16LinearMethod)
    assert isinstance(get_linear_method(QuantConfig.fp8_block()), W8A16LinearMethod)
    assert isinstance(get_linear_method(QuantConfig.smoothquant_per_channel()), SmoothQuantLinearMethod)
    assert isinstance(get_moe_method(QuantConfig.smoothquant_per_channel()), W8A16MoeMethod)
    with pytest.raises(ValueError, match="not supported for MoE"):
        get_moe_method(QuantConfig.int4_groupwise())
    lin = LinearBase(512, 256, bias=True, quant=QuantConfig.int4_groupwise(128))
    assert lin.weight.shape == (256, 64) and lin.weight.dtype == torch.int32
    assert lin.weight_scale.shape == (256, 4) and lin.weight_zeros.dtype == torch.float32
    lin8 = LinearBase(512, 256, quant=QuantConfig.fp8_block())
    assert lin8.weight.dtype == torch.uint8 and lin8.weight_scale_inv.shape == (2, 4)
    q = QuantConfig.for_runtime_scheme("int8-blockwise")
    assert q.group_k == 128 and q.storage_dtype == torch.int8
    with pytest.raises(ValueError):
        QuantConfig.for_runtime_scheme("int3")


def test_quantisers_match_oracle_bit_exactly():
    from lite_llama_amd.quantization import (quantize_fp8_per_channel, quantize_int4_groupwise,
                                             quantize_int8_groupwise, quantize_int8_per_channel)
    from oracle import oracle as O

    w = torch.randn(16, 256) * 0.05
    for mine, ref in [(quantize_int4_groupwise(w, 128), O.quantize_int4_groupwise(w, 128)),
                      (quantize_int8_per_channel(w), O.quantize_int8_per_channel(w)),
                      (quantize_int8_groupwise(w, 128), O.quantize_int8_groupwise(w, 128)),
                      (quantize_fp8_per_channel(w), O.quantize_fp8_per_channel(w))]:
        for a, b in zip(mine, ref):
            assert torch.equal(a, b)


def test_model_skeleton_and_tp_rules():
    from lite_llama_amd.model import GEOMETRY, CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    m = CausalLM(tiny_geometry(), QuantConfig.int4_groupwise(128))
    names = dict(m.named_parameters())
    assert "layers.0.self_attn.kv_proj.weight_zeros" in names and "layers.1.mlp.down_proj.weight_scale" in names
    g = GEOMETRY["qwen2.5-7b"]
    assert (g.q_size, g.kv_size, g.intermediate_size) == (3584, 512, 18944)
    moe = CausalLM(tiny_geometry(num_experts=4, num_experts_per_tok=2, moe_intermediate_size=128))
    assert moe.layers[0].mlp.experts["gate_up_proj"].shape == (4, 256, 256)


def test_decode_attention_shape_gate():
    """Which shapes the one-launch decode attention accepts is decided on the host, from shapes, strides
    and dtypes only (no launch): head_dim >= 64, <= 16 query heads per KV head, dense head rows, 16-bit
    tables of the activation dtype."""
    from lite_llama_amd.kernels.attention import decode_attention_supported as ok

    def qkv(hq, hkv, d, dt=torch.float16):
        fused = torch.zeros(3, (hq + 2 * hkv) * d, dtype=dt)
        return fused[:, : hq * d].view(3, hq, d), fused[:, hq * d:].view(3, 2 * hkv, d)

    cos = torch.zeros(16, 128, dtype=torch.float16)
    q, kv = qkv(28, 4, 128)
    assert ok(q, kv, cos, 4)
    assert not ok(*qkv(28, 4, 32), cos, 4)                      # rotation partner would sit in another lane
    assert not ok(*qkv(34, 2, 64), cos, 2)                      # 17 query heads per KV head: two head groups
    assert not ok(q, kv, cos.float(), 4)                        # fp32 tables
    assert not ok(q.transpose(0, 1), kv, cos, 4)                # head rows not dense
    qb, kvb = qkv(8, 2, 64, torch.bfloat16)
    assert ok(qb, kvb, cos.bfloat16(), 2) and not ok(qb, kvb, cos, 2)


def test_oneshot_all_reduce_needs_a_tensor_parallel_group():
    """The opt-in peer-to-peer all-reduce is a property of an initialised TP group of 2..8 ranks; without one the
    request is refused and ``all_reduce_tp`` stays the identity / the backend's collective."""
    import pytest
    import torch

    from lite_llama_amd.distributed import parallel_state as ps

    with pytest.raises(RuntimeError, match="tensor-parallel group"):
        ps.enable_oneshot_all_reduce(64)
    assert ps.oneshot_error() == 0
    x = torch.ones(8)
    assert ps.all_reduce_tp(x) is x


def test_xcd_aware_slice_map_is_a_bijection():
    """Restatement of gemm_w4_v3.hip's XCD-aware map of the split-K partial launches (V3Params::xcd_shift): workgroup
    b = 8 q + x takes k-slice x % gt of tile q * (8 / gt) + x / gt.  For every power-of-two split up to 8 and every tile count
    with tiles * gt % 8 == 0 the map is one-to-one onto (tile, slice), and a workgroup's slice depends on b % 8 only -- the
    XCD it runs on under the round-robin dispatch -- so one XCD's L2 only ever sees one slice of the activation matrix."""
    for sh in range(4):
        gt = 1 << sh
        for tiles in range(1, 80):
            if (tiles * gt) % 8:
                continue
            seen = set()
            for b in range(tiles * gt):
                x, q = b & 7, b >> 3
                j, tile = x & (gt - 1), q * (8 >> sh) + (x >> sh)
                assert 0 <= tile < tiles and j == (b % 8) % gt
                seen.add((tile, j))
            assert len(seen) == tiles * gt


def _v3_plan(lib, n, k, epilogue):
    import ctypes

    out = (ctypes.c_int32 * 16)()
    assert lib.ll_w4a16_v3_plan(64, n, k, 128, epilogue, out) == 0
    keys = ["grid", "nf", "tiles", "chunks", "slots", "gt", "gbase", "grem", "glead", "xcd_shift", "upw", "r0", "r1", "r2", "r3", "cus"]
    return dict(zip(keys, list(out)))


def _v3_segments(pl, b):
    """Restatement of wgemm3_kernel's range decode (gemm_w4_v3.hip): the (tile, first chunk, last chunk) segments workgroup b
    walks, in execution order (tail of the last tile, whole tiles, head of the first tile)."""
    chunks = pl["chunks"]
    if pl["gt"]:
        if pl["xcd_shift"] >= 0:
            x, q = b & 7, b >> 3
            j, gtile = x & (pl["gt"] - 1), q * (8 >> pl["xcd_shift"]) + (x >> pl["xcd_shift"])
        else:
            gtile, j = divmod(b, pl["gt"])
        lo = j * pl["gbase"] + min(j, pl["grem"])
        ub = gtile * chunks + lo
        ue = (gtile + 1) * chunks if j == pl["gt"] - 1 else ub + pl["gbase"] + (1 if j < pl["grem"] else 0)
    else:
        ub, ue = b * pl["upw"], min((b + 1) * pl["upw"], pl["tiles"] * chunks)
    if ub >= ue:
        return []
    (tA, cA), (tZ, cZ) = divmod(ub, chunks), divmod(ue - 1, chunks)
    if tA == tZ:
        return [(tA, cA, cZ)]
    segs = []
    if cZ != chunks - 1:
        segs.append((tZ, 0, cZ))
    tF = tA + (1 if cA != 0 else 0)
    tL = tZ - (1 if cZ != chunks - 1 else 0)
    segs += [(t, 0, chunks - 1) for t in range(tF, tL + 1)]
    if cA != 0:
        segs.append((tA, cA, chunks - 1))
    return segs


def test_w4a16_plans_cover_every_unit_once():
    """Every launch plan of the pre-packed engine -- tile groups with an owner lead, the XCD-aware slice map of the split-K
    partial launches, stream-K -- hands every (tile, chunk) unit to exactly one workgroup (the kernel's per-workgroup range
    decode restated on the host plan, ``ll_w4a16_v3_plan``); a tile has at most ``slots`` contributing workgroups and every
    contributor of a tile has a lower workgroup number than its owner (the merge waits point downwards only)."""
    from lite_llama_amd import _lib

    lib = _lib.lib()
    shapes = [(4608, 3584, 2), (3584, 3584, 2), (3584, 18944, 2), (37888, 3584, 1), (37888, 3584, 0), (4608, 3584, 0),
              (152064, 3584, 0), (2048, 1536, 0), (17920, 1536, 1)]
    shapes += [(128 * t, 128 * c, 0) for t in (1, 3, 28, 131, 160, 255, 300) for c in (1, 4, 7, 28, 37, 148)]
    shapes += [(256 * t, 128 * c, 1) for t in (128, 148, 191, 255) for c in (5, 28, 33)]
    for n, k, ep in shapes:
        pl = _v3_plan(lib, n, k, ep)
        assert pl["tiles"] * pl["nf"] * 128 == n and pl["chunks"] * 128 == k
        seen = {}
        for b in range(pl["grid"]):
            for t, lo, hi in _v3_segments(pl, b):
                assert 0 <= t < pl["tiles"] and 0 <= lo <= hi < pl["chunks"], (n, k, ep, b, t, lo, hi)
                for c in range(lo, hi + 1):
                    assert (t, c) not in seen, (n, k, ep, b, t, c)
                    seen[(t, c)] = b
        assert len(seen) == pl["tiles"] * pl["chunks"], (n, k, ep, pl)
        for t in range(pl["tiles"]):
            owners = {seen[(t, pl["chunks"] - 1)]}
            wgs = sorted({seen[(t, c)] for c in range(pl["chunks"])})
            assert len(wgs) <= max(pl["slots"], 1), (n, k, ep, t, wgs, pl)
            if ep != 2 and pl["xcd_shift"] < 0:
                assert max(wgs) in owners, (n, k, ep, t, wgs)   # the owner is the highest-numbered workgroup of its tile
    pl = _v3_plan(lib, 37888, 3584, 1)   # the headline gate|up launch: stream-K over 256-row tiles
    assert (pl["nf"], pl["tiles"], pl["chunks"], pl["upw"], pl["grid"]) == (2, 148, 28, 17, 244), pl


def test_row_group_engine_plan_assigns_every_row_group_once():
    """The round-5 row-group engine (gemm_w4_v4.hip) for finished-output launches: workgroup t owns 32-row groups
    ``[t * rbase + min(t, rrem), + rbase + (t < rrem))`` and all of K -- every group exactly once, each workgroup within one of the
    template bound, the headline gate|up as 160 workgroups of five groups + 96 of four; shapes it does not serve stay on the unit loop."""
    import ctypes
    from lite_llama_amd import _lib

    lib = _lib.lib()

    def plan(n, k, ep):
        out = (ctypes.c_int32 * 8)()
        assert lib.ll_w4a16_v4_plan(64, n, k, 128, ep, ctypes.cast(out, ctypes.c_void_p)) == 0
        return list(out)

    takes, grid, nrgt, rbase, rrem, cus, lds, _ = plan(37888, 3584, 1)
    assert (takes, grid, nrgt, rbase, rrem) == (1, 256, 5, 4, 160) and lds <= 160 * 1024
    for n in (37888, 28672, 40960, 33024, 32768 + 128):
        takes, grid, nrgt, rbase, rrem, cus, lds, _ = plan(n, 512, 0)
        if not takes:
            continue
        owned = []
        for t in range(grid):
            lo = t * rbase + min(t, rrem)
            cnt = rbase + (1 if t < rrem else 0)
            assert nrgt - 1 <= cnt <= nrgt, (n, t, cnt, nrgt)
            owned += list(range(lo, lo + cnt))
        assert owned == list(range(n // 32)), n
    assert plan(4608, 3584, 0)[0] == 0 and plan(3584, 18944, 0)[0] == 0   # too few row groups per CU: the unit loop
    assert plan(37888, 3584, 2)[0] == 0                                    # split-K partial launches: unit loop by default
    assert plan(37888, 3584, 1 | (2 << 8))[0] == 0                         # a forced unit-loop tile width (tests / tuning)


def test_int8_rows_regroup_only_their_leading_dimensions():
    """kernels/norm_act.py::Int8Rows stands in for the fp16 activations of a smoothquant block (the quantiser ran inside the
    norm launch): callers re-group its leading dimensions like a tensor's; the row width never changes."""
    import pytest
    from lite_llama_amd.kernels.norm_act import Int8Rows

    q = torch.zeros(6, 32, dtype=torch.int8)
    s = torch.ones(6)
    r = Int8Rows(q, s, (2, 3, 32))
    assert r.shape == (2, 3, 32) and r.dtype == torch.float16 and r.numel() == 6 * 32
    flat = r.view(-1, 32)
    assert flat.shape == (6, 32) and flat.q is q and flat.scale is s
    assert r.view(3, -1, 32).shape == (3, 2, 32) and r.view((6, 32)).shape == (6, 32)
    with pytest.raises(ValueError):
        r.view(-1, 16)


def test_scaled_int32_partials_materialise_is_the_reference_smoothquant_epilogue():
    """kernels/norm_act.py::ScaledInt32Partials (what a smoothquant projection leaves for its consumer at decode shapes): summing
    the exact int32 split-K planes and applying (acc.f32 * a_scale[m]) * w_scale[n] (+ bias) -- w8a8.py:118-120 -- gives the
    oracle's smoothquant_matmul bit for bit, however the contraction was split."""
    from lite_llama_amd.kernels.norm_act import ScaledInt32Partials
    from oracle import oracle as O

    torch.manual_seed(3)
    M, N, K_ = 5, 48, 256
    x = (torch.randn(M, K_) * 0.5).half()
    qw, sc = O.quantize_int8_per_channel(torch.randn(N, K_) * 0.05)
    bias = (torch.randn(N) * 0.1).half()
    qa, a_scale = O.quantize_activations_int8(x)
    cuts = [0, 64, 192, 256]
    planes = torch.stack([(qa[:, a:b].int() @ qw[:, a:b].int().T) for a, b in zip(cuts[:-1], cuts[1:])]).to(torch.int32)
    acc_ref, _, _ = O.smoothquant_int32_acc(x, qw)
    assert torch.equal(planes.sum(0, dtype=torch.int32), acc_ref)
    for b in (None, bias):
        parts = ScaledInt32Partials(planes, (M, N), a_scale, sc.reshape(-1).float(), b)
        assert torch.equal(parts.materialise(), O.smoothquant_matmul(x, qw, sc, bias=b))
