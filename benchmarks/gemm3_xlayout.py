"""Round-3 experiment: does the activation tile's address pattern bound the W4A16 decode GEMM?

The [64 x K] fp16 activation matrix has a row stride of K * 2 bytes (7168 = 56 lines, 37888 = 296 lines): the 128 lines
of one 128-k chunk tile then fall on few L2 channels if the channel is picked from low line-address bits.  This times
the four GEMM launches of a Qwen2.5-7B layer (the step's own forms: q|k|v and o / down as split-K partials, gate|up with
the swiglu epilogue) with the activation rows padded by PAD halves, or chunk-major (LL_GEMM3_XCM=1, timing only).
Weights rotate over ~700 MB of copies (COPIES=1: one copy, i.e. L2 / MALL-warm).  One JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
M = int(os.environ.get("M", 64))
pads = [int(p) for p in os.environ.get("PADS", "0,64").split(",")]
shapes = [("qkv", 4608, 3584, 2), ("o", 3584, 3584, 2), ("gateup", 37888, 3584, 1), ("down", 3584, 18944, 2)]
if os.environ.get("ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["ONLY"].split(",")]
res = {"lib": os.environ.get("LL_LIB_OVERRIDE", "default"), "xcm": bool(os.environ.get("LL_GEMM3_XCM")), "copies": os.environ.get("COPIES", "rot")}
for name, n, k, mode in shapes:
    wbytes = n * k // 2 + n * (k // 128) * 8
    copies = int(os.environ.get("COPIES", 0)) or max(2, int(700e6 // wbytes))
    pw, ps = [], []
    for _ in range(copies):
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
        sc = torch.rand(n, k // 128, device=dev) * 0.01 + 0.005
        zr = torch.randint(0, 16, (n, k // 128), device=dev).float()
        pw.append(Q.pack_w4a16_weights(qw)); ps.append(Q.pack_w4a16_scales(sc, zr))
        del qw, sc, zr
    for pad in pads:
        buf = torch.randn(M, k + pad, device=dev, dtype=torch.float16)
        x = buf[:, :k]

        def call(i):
            if mode == 1:
                return Q.w4a16_matmul_prepacked(x, pw[i], ps[i], group_size=128, gate_up_swiglu=True)
            return Q.w4a16_matmul_partials(x, pw[i], ps[i], group_size=128)

        call(0); torch.cuda.synchronize()
        reps = max(copies, 16)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                call(i % copies)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
        res[f"{name}:pad{pad}"] = round(best, 2)
        print(f"{name:7s} pad={pad:5d} {best:7.2f} us  {wbytes / best / 1e6:5.2f} TB/s", flush=True)
        del g
    del pw, ps
    torch.cuda.empty_cache()
print(json.dumps(res))
