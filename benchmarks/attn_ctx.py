"""flash_decoding time vs context length (Qwen2.5-7B decode shape: batch 64, 28 / 4 heads of 128), hipGraph replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K

dev = "cuda"
B, HQ, HKV, D = int(os.environ.get("B", 64)), int(os.environ.get("HQ", 28)), int(os.environ.get("HKV", 4)), 128
for ctx in [int(c) for c in os.environ.get("CTX", "64,128,256,384,512,640,1024,2048").split(",")]:
    rows = B * ctx
    pools = [torch.randn(rows, 2 * HKV, D, device=dev, dtype=torch.float16) * 0.5 for _ in range(8)]
    q = torch.randn(B, HQ, D, device=dev, dtype=torch.float16) * 0.3
    table = torch.randperm(rows, device=dev).int().view(B, ctx)
    req = torch.arange(B, dtype=torch.int32, device=dev)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    if os.environ.get("FP8"):  # e4m3 pool (extension): half the K/V bytes
        pools = [p.to(torch.float8_e4m3fn) for p in pools]
        f = lambda p: K.flash_decoding_fp8kv(q, p[:, :HKV], p[:, HKV:], 1.0 / D**0.5, table, req, seq, ctx)
    else:
        f = lambda p: K.flash_decoding(q, p[:, :HKV], p[:, HKV:], 1.0 / D**0.5, table, req, seq, ctx)
    f(pools[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p in pools:
            f(p)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * len(pools))
    byt = rows * 2 * HKV * D * (1 if os.environ.get("FP8") else 2)
    print(f"ctx {ctx:5d}: {us:7.2f} us  {byt / us / 1e6:5.2f} TB/s  ({byt / 1e6:.1f} MB, {B * HKV * ((ctx + 127) // 128)} waves)", flush=True)
