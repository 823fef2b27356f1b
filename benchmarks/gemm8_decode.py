"""Decode-shaped 8-bit GEMMs (Llama-3-8B W8A8 at batch 32; Qwen3-30B-A3B fp8 attention projections at batch 64):
us per launch and weight bytes / time, hipGraph-replayed over rotating weight copies.  LL_DENSE8_OFF=1 -> generic engine."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
res = {}
def run(name, m, n, k, kind):
    copies = max(2, int(400e6 // (n * k)))
    x = (torch.randn(m, k, device=dev, generator=g) * 0.5).half()
    if kind == "w8a8":
        ws = [torch.randint(-127, 128, (n, k), device=dev, dtype=torch.int8, generator=g) for _ in range(copies)]
        sc = torch.rand(n, device=dev, generator=g) * 0.01 + 0.001
        f = lambda i: K.smoothquant_matmul(x, ws[i], sc)
    else:
        ws = [(torch.randint(0, 256, (n, k), device=dev, dtype=torch.uint8, generator=g) & 0x77) for _ in range(copies)]
        sc = torch.rand((n + 127) // 128, (k + 127) // 128, device=dev, generator=g) * 0.01 + 0.001
        f = lambda i: K.w8a16_matmul(x, ws[i], sc, group_n=128, group_k=128)
    f(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(copies):
            f(i)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * copies)
    res[name] = round(us, 2)
    print(f"{name:28s} M={m:2d} N={n:6d} K={k:6d}: {us:7.2f} us  {n * k / us / 1e6:5.2f} TB/s", flush=True)
for name, n, k in (("llama3 qkv", 6144, 4096), ("llama3 o", 4096, 4096), ("llama3 gate|up", 28672, 4096), ("llama3 down", 4096, 14336)):
    run("w8a8 " + name, 32, n, k, "w8a8")
for name, n, k in (("qwen3-moe qkv", 5120, 2048), ("qwen3-moe o", 2048, 4096)):
    run("fp8 " + name, 64, n, k, "fp8")
print(json.dumps(res))
