"""GPU time per W4A16 GEMM shape of Qwen2.5-7B (hipGraph-replayed, rotating weight copies so the
stream really comes from HBM).  Prints us/launch and algorithmic TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K

dev = "cuda"
M = int(os.environ.get("M", 64))
shapes = [("q/o", 3584, 3584), ("kv", 1024, 3584), ("gate/up", 18944, 3584), ("down", 3584, 18944)]
if os.environ.get("SHAPES"):
    shapes = [(f"s{i}", *map(int, t.split("x"))) for i, t in enumerate(os.environ["SHAPES"].split(","))]
for name, n, k in shapes:
    wbytes = n * k // 2 + 2 * n * (k // 128) * 4
    copies = max(2, int(700e6 // wbytes))
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
           torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(copies)]
    x = torch.randn(M, k, device=dev, dtype=torch.float16)
    packed = [None] * copies
    if os.environ.get("PACKED", "1") == "1":
        from lite_llama_amd.kernels.quantization import pack_w4a16_scales
        packed = [pack_w4a16_scales(w[1], w[2]) for w in ws]
    K.w4a16_matmul(x, *ws[0], group_size=128, packed_scales=packed[0])
    torch.cuda.synchronize()
    reps = max(copies, 16)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            K.w4a16_matmul(x, *ws[i % copies], group_size=128, packed_scales=packed[i % copies])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    print(f"{name:8s} N={n:6d} K={k:6d} M={M}: {us:7.2f} us/launch  {wbytes / us / 1e6:6.2f} TB/s  ({wbytes/1e6:.1f} MB)", flush=True)
