"""The short-stream engine's all-of-K form (csrc/gemm_short_full.hip: finished fp16 outputs) against the unit loop at the shapes it
serves: the reference-shaped layer's q / k|v / o projections (Qwen2.5-7B, Llama-3-8B) and TP-shard fused gate|up.  us per launch over
rotating weights (hipGraph replays), the engine's plan.  LL_GEMM_SF=0 in the environment times the unit loop instead (read once)."""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q
from lite_llama_amd import _lib as L

dev, M = "cuda", int(os.environ.get("M", 64))
SHAPES = {"q/o 3584x3584": (3584, 3584, False), "k|v 1024x3584": (1024, 3584, False), "gate|up tp8 4608x3584": (4608, 3584, True),
          "gate|up tp4 9472x3584": (9472, 3584, True), "llama q/o 4096x4096": (4096, 4096, False), "llama k|v 2048x4096": (2048, 4096, False)}
res = {"m": M, "env": {k: v for k, v in os.environ.items() if k.startswith("LL_GEMM")}}
for name, (n, k, swiglu) in SHAPES.items():
    out = (ctypes.c_int32 * 8)()
    L.lib().ll_w4a16_short_full_plan(M, n, k, 128, out)
    torch.manual_seed(1)
    wbytes = n * k // 2 + n * (k // 128) * 8
    copies = max(2, int(500e6 // wbytes))
    x = (torch.randn(M, k, device=dev) * 0.5).half()
    bias = None if swiglu else (torch.randn(n, device=dev) * 0.1).half()
    pw, ps = [], []
    for c in range(copies):
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
        sc = torch.rand(n, k // 128, device=dev) * 0.01 + 0.005
        zr = torch.randint(0, 16, (n, k // 128), device=dev).float()
        pw.append(Q.pack_w4a16_weights(qw)); ps.append(Q.pack_w4a16_scales(sc, zr))
        del qw, sc, zr
    kw = dict(gate_up_swiglu=True) if swiglu else dict(bias=bias)
    Q.w4a16_matmul_prepacked(x, pw[0], ps[0], **kw)
    reps = max(copies, 16)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            Q.w4a16_matmul_prepacked(x, pw[i % copies], ps[i % copies], **kw)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
    res[name] = {"us": round(best, 2), "MB": round(wbytes / 1e6, 1), "plan[takes,items,R,MT,halves,P,slots,lds]": list(out)}
    del pw, ps
    torch.cuda.empty_cache()
print(json.dumps(res))
