"""Round 5: the row-group engine (gemm_w4_v4.hip) against the unit-loop engine (gemm_w4_v3.hip) and an fp32 reference --
finished outputs (bias / fused swiglu) and split-K planes, at the headline shapes and at odd batch sizes.  One JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
torch.manual_seed(0)
res = {}
ok = True


def dequant(qw, sc, zr, g=128):
    n, kp = qw.shape
    sh = torch.arange(8, device=qw.device, dtype=torch.int32) * 4
    nib = ((qw.unsqueeze(-1) >> sh) & 15).reshape(n, kp * 8).float()
    return ((nib - zr.repeat_interleave(g, 1)) * sc.repeat_interleave(g, 1))


SHAPES = [("gateup", 37888, 3584), ("down", 3584, 18944), ("o", 3584, 3584), ("qkv", 4608, 3584), ("small", 1024, 512),
          ("gu_1p5b", 17920, 1536)]
if os.environ.get("EXTRA"):  # with LL_GEMM4_MINFILL=1: finished outputs at 2 / 3-4 / 4-5 row groups per workgroup
    SHAPES = [("n12288", 12288, 1024), ("llama_gu", 28672, 4096), ("n40960", 40960, 512), ("n33024", 33024, 256)]
for name, n, k in SHAPES:
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
    sc = torch.rand(n, k // 128, device=dev) * 0.01 + 0.005
    zr = torch.randint(0, 16, (n, k // 128), device=dev).float()
    pw, ps = Q.pack_w4a16_weights(qw), Q.pack_w4a16_scales(sc, zr)
    wd = dequant(qw, sc, zr)
    bias = (torch.randn(n, device=dev) * 0.1).half()
    for m in (64, 33, 17, 1):
        x = (torch.randn(m, k, device=dev) * 0.5).half()
        ref = x.float() @ wd.t()
        tol = 2e-2 * ref.abs().max().item()
        # planes
        p4 = Q.w4a16_matmul_partials(x, pw, ps)
        p3 = Q.w4a16_matmul_partials(x, pw, ps, _unit_loop_engine=True)
        if p4 is not None:
            s4, s3 = p4.parts.sum(0), p3.parts.sum(0)
            e4, e3 = (s4 - ref).abs().max().item(), (s3 - ref).abs().max().item()
            d43 = (s4 - s3).abs().max().item()
            good = e4 <= tol and p4.parts.shape == p3.parts.shape
            ok &= good
            res[f"{name}:m{m}:planes"] = {"S": p4.parts.shape[0], "err_v4": round(e4, 5), "err_v3": round(e3, 5), "v4_v3": round(d43, 6), "tol": round(tol, 4), "ok": good}
        # finished outputs (bias) and the fused swiglu
        for mode in ("bias", "swiglu"):
            if mode == "swiglu":
                y4 = Q.w4a16_matmul_prepacked(x, pw, ps, gate_up_swiglu=True)
                y3 = Q.w4a16_matmul_prepacked(x, pw, ps, gate_up_swiglu=True, _tile_blocks=2 if n % 256 == 0 else 1)
                r16 = ref.half().float()
                g_, u_ = r16[:, 0::2], r16[:, 1::2]
                rr = torch.nn.functional.silu(g_) * u_
            else:
                y4 = Q.w4a16_matmul_prepacked(x, pw, ps, bias=bias)
                y3 = Q.w4a16_matmul_prepacked(x, pw, ps, bias=bias, _tile_blocks=1)
                rr = ref + bias.float()
            t2 = 2e-2 * rr.abs().max().item()
            e4, e3 = (y4.float() - rr).abs().max().item(), (y3.float() - rr).abs().max().item()
            good = e4 <= t2 and bool(torch.isfinite(y4).all())
            ok &= good
            res[f"{name}:m{m}:{mode}"] = {"err_v4": round(e4, 5), "err_v3": round(e3, 5), "bit_equal_v3": bool(torch.equal(y4, y3)),
                                          "max_v4_v3": round((y4.float() - y3.float()).abs().max().item(), 5), "tol": round(t2, 4), "ok": good}
    # repeatability
    a = Q.w4a16_matmul_prepacked(x, pw, ps, gate_up_swiglu=True)
    b = Q.w4a16_matmul_prepacked(x, pw, ps, gate_up_swiglu=True)
    ok &= bool(torch.equal(a, b))
    del qw, sc, zr, pw, ps, wd
    torch.cuda.empty_cache()
res["ALL_OK"] = bool(ok)
bad = {k_: v for k_, v in res.items() if isinstance(v, dict) and not v["ok"]}
print(json.dumps({"ALL_OK": bool(ok), "bad": bad}))
print(json.dumps(res))
