"""Short-stream engine for the row-major weight formats (csrc/gemm_short_dense.hip) against the streaming split-K engine
(gemm_w8_skinny.hip): split-K partial launches at the dense projection shapes of Qwen3-30B-A3B (fp8 128 x 128 blocks), of a 1.5B
bf16 model and of TP shards.  us per launch (hipGraph replays over ~600 MB of rotating weights), planes, max error vs fp32.
The engine is chosen once per process: run once as is and once with LL_DENSE_SS=0.  One JSON line.

    python benchmarks/gemm_short_dense.py ; LL_DENSE_SS=0 python benchmarks/gemm_short_dense.py
"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
SHAPES = {  # name: (fmt, m, n, k)
    "c5_qkv_fp8": ("fp8", 64, 5120, 2048), "c5_o_fp8": ("fp8", 64, 2048, 4096),
    "c2_qkv_bf16": ("bf16", 32, 2048, 1536), "c2_o_bf16": ("bf16", 32, 1536, 1536),
    "l3_o_int8_tp4": ("int8", 32, 4096, 1024), "q7_o_f16": ("f16", 64, 3584, 3584 // 4),
}
names = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
res = {"ss": os.environ.get("LL_DENSE_SS", "1") != "0"}


def timed(fn, copies):
    fn(0); torch.cuda.synchronize()
    reps = max(copies, 16)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i % copies)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
    return best


for name in names:
    fmt, m, n, k = SHAPES[name]
    torch.manual_seed(1)
    eb = 2 if fmt in ("f16", "bf16") else 1
    copies = max(2, int(600e6 // (n * k * eb)))
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    x = (torch.randn(m, k, device=dev) * 0.5).to(dt)
    ws, scs = [], []
    for c in range(copies):
        if eb == 2:
            ws.append((torch.randn(n, k, device=dev) * 0.05).to(dt)); scs.append(None)
        elif fmt == "fp8":
            q = torch.randint(0, 256, (n, k), device=dev, dtype=torch.int64).to(torch.uint8)
            ws.append(torch.where((q & 0x7F) == 0x7F, q - 1, q)); scs.append(torch.rand(n // 128, k // 128, device=dev) * 2e-4 + 1e-4)
        else:
            ws.append(torch.randint(-127, 128, (n, k), device=dev, dtype=torch.int8)); scs.append(torch.rand(n, 1, device=dev) * 1e-3 + 5e-4)
    kw = {} if eb == 2 else ({"group_n": 128, "group_k": 128} if fmt == "fp8" else {"group_n": 1, "group_k": k})
    call = lambda i: Q.dense_matmul_partials(x, ws[i], scs[i], max_splits=8, **kw)
    p = call(0)
    if p is None:
        res[name] = "declined"
        continue
    if eb == 2:
        ref = x.float() @ ws[0].float().T
    else:
        import oracle.oracle as O
        ref = O.w8a16_matmul(x.cpu(), ws[0].cpu(), scs[0].cpu(), **kw).float().to(dev)
    err = ((p.parts.sum(0) - ref).abs().max() / ref.abs().max()).item()
    res[name] = {"planes": int(p.parts.shape[0]), "rel_err": round(err, 5), "us": round(timed(call, copies), 2), "MB": round(n * k * eb / 1e6, 1)}
    del ws, scs
    torch.cuda.empty_cache()
print(json.dumps(res))
