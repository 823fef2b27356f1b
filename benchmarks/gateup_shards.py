"""The fused gate|up + swiglu launch at TP-shard widths (Qwen2.5-7B: 2 x 18944 / tp rows, K = 3584, M = 64): us per launch over
rotating weights (hipGraph replays), max error against an fp32 reference.  The engine is chosen by the environment, read once
per process: default = unit loop below 3 row groups per CU; LL_GEMM4_SMALL=1 = the row-group engine at any fill."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev, M, K = "cuda", int(os.environ.get("M", 64)), 3584
res = {"env": {k: v for k, v in os.environ.items() if k.startswith("LL_GEMM")}}
for tp in (1, 2, 4, 8):
    inter = {1: 18944, 2: 9472, 4: 4736, 8: 2304}[tp]   # (tp 8: the extension plan's 2304 / 2432 channels per rank)
    n = 2 * inter
    torch.manual_seed(tp)
    wbytes = n * K // 2 + n * (K // 128) * 8
    copies = max(2, int(600e6 // wbytes))
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    pw, ps, ref = [], [], None
    for c in range(copies):
        qw = torch.randint(-(2**31), 2**31 - 1, (n, K // 8), dtype=torch.int64, device=dev).to(torch.int32)
        sc = torch.rand(n, K // 128, device=dev) * 0.01 + 0.005
        zr = torch.randint(0, 16, (n, K // 128), device=dev).float()
        pw.append(Q.pack_w4a16_weights(qw)); ps.append(Q.pack_w4a16_scales(sc, zr))
        if c == 0:
            nib = torch.stack([(qw >> (4 * j)) & 15 for j in range(8)], dim=-1).reshape(n, K).float()
            wd = (nib - zr.repeat_interleave(128, dim=1)) * sc.repeat_interleave(128, dim=1)
            r16 = (x.float() @ wd.t()).half().float()
            ref = torch.nn.functional.silu(r16[:, 0::2]) * r16[:, 1::2]
        del qw, sc, zr
    y = Q.w4a16_matmul_prepacked(x, pw[0], ps[0], gate_up_swiglu=True)
    err = (y.float() - ref).abs().max().item()
    again = Q.w4a16_matmul_prepacked(x, pw[0], ps[0], gate_up_swiglu=True)
    reps = max(copies, 16)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            Q.w4a16_matmul_prepacked(x, pw[i % copies], ps[i % copies], gate_up_swiglu=True)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
    res[f"tp{tp}"] = {"n": n, "MB": round(wbytes / 1e6, 1), "us": round(best, 2), "TBps": round(wbytes / best / 1e6, 2),
                      "max_err": round(err, 5), "tol": round(2e-2 * ref.abs().max().item(), 4), "repeatable": bool(torch.equal(y, again))}
    del pw, ps
    torch.cuda.empty_cache()
print(json.dumps(res))
