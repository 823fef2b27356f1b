"""o / down projection as split-K partials + the add-and-normalise that sums them (the step's launch pair), hipGraph replays over
rotating weights: us per (GEMM, norm) pair and for the norm alone.  Knobs are the library's (LL_GEMM3_XCD / LL_GEMM3_FILL)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q
from lite_llama_amd.kernels.norm_act import skip_rmsnorm_partials

dev = "cuda"
res = {"xcd": os.environ.get("LL_GEMM3_XCD", "8"), "fill": os.environ.get("LL_GEMM3_FILL", "85")}
for name, n, k in [("o", 3584, 3584), ("down", 3584, 18944)]:
    wbytes = n * k // 2 + n * (k // 128) * 8
    copies = max(2, int(400e6 // wbytes))
    pw, ps = [], []
    for _ in range(copies):
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
        pw.append(Q.pack_w4a16_weights(qw))
        ps.append(Q.pack_w4a16_scales(torch.rand(n, k // 128, device=dev) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=dev).float()))
    x = torch.randn(64, k, device=dev, dtype=torch.float16)
    res_t = torch.randn(64, n, device=dev, dtype=torch.float16)
    nw = torch.ones(n, device=dev, dtype=torch.float16)

    def pair(i, with_norm=True, with_gemm=True, parts=[None]):
        if with_gemm:
            parts[0] = Q.w4a16_matmul_partials(x, pw[i], ps[i], group_size=128)
        if with_norm:
            skip_rmsnorm_partials(parts[0], res_t, nw, 1e-6)

    pair(0); torch.cuda.synchronize()
    res[name + ":planes"] = int(pair.__defaults__[2][0].parts.shape[0])
    for label, kw in (("pair", {}), ("gemm", {"with_norm": False}), ("norm", {"with_gemm": False})):
        reps = max(copies, 16)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                pair(i % copies, **kw)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
        res[f"{name}:{label}"] = round(best, 2)
        del g
print(json.dumps(res))
