"""GPU time per W4A16 GEMM launch of the Qwen2.5-7B decode step, both decode engines side by side
(reference-format v2 vs pre-packed v3), hipGraph-replayed over rotating weight copies so the stream
really comes from HBM.  Prints us/launch and algorithmic TB/s; JSON line at the end."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
M = int(os.environ.get("M", 64))
shapes = [("qkv", 4608, 3584, 0), ("o", 3584, 3584, 0), ("gate|up", 37888, 3584, 1), ("down", 3584, 18944, 0)]
if os.environ.get("SHAPES"):
    shapes = [(f"s{i}", *map(int, t.split("x")), 0) for i, t in enumerate(os.environ["SHAPES"].split(","))]
engines = os.environ.get("ENGINES", "v2,v3").split(",")
res = {}
for name, n, k, epi in shapes:
    wbytes = n * k // 2 + n * (k // 128) * 8
    copies = max(2, int(700e6 // wbytes))
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
           torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(copies)]
    x = torch.randn(M, k, device=dev, dtype=torch.float16)
    ps = [Q.pack_w4a16_scales(w[1], w[2]) for w in ws]
    pw = [Q.pack_w4a16_weights(w[0]) for w in ws]

    def call(eng, i):
        if eng.startswith("v3"):  # v3 = host plan, v3a / v3b = 128- / 256-row tiles forced
            tb = {"v3": 0, "v3a": 1, "v3b": 2}[eng]
            return Q.w4a16_matmul_prepacked(x, pw[i], ps[i], group_size=128, gate_up_swiglu=bool(epi), _tile_blocks=tb)
        if epi:
            return Q.w4a16_gate_up_swiglu(x, *ws[i], group_size=128, packed_scales=ps[i])
        return K.w4a16_matmul(x, *ws[i], group_size=128, packed_scales=ps[i])

    for eng in engines:
        call(eng, 0)
        torch.cuda.synchronize()
        reps = max(copies, 16)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                call(eng, i % copies)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
        res[f"{name}:{eng}"] = round(us, 2)
        print(f"{name:8s} {eng} N={n:6d} K={k:6d} M={M}: {us:7.2f} us/launch  {wbytes / us / 1e6:6.2f} TB/s  ({wbytes/1e6:.1f} MB)", flush=True)
    del ws, ps, pw
    torch.cuda.empty_cache()
print(json.dumps(res))
