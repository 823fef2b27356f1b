"""In-kernel timeline of wss_kernel (csrc/gemm_short.hip; needs a -DSS_TIMELINE build:
    python tools/build_variant.py ss_tl gemm_short.hip -DSS_TIMELINE;  LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/ss_tl.so python benchmarks/gemm_short_timeline.py).
Stamps are s_memrealtime (100 MHz, one counter for the chip) per (workgroup, wave): printed in us after the launch's first stamp,
as median / min / max over workgroups.  Consumers: 0 entry, 1 weight loads issued, 2 barrier A passed (first half of the activation
slice in LDS), 3 + 2 i piece i landed, 4 + 2 i piece i multiplied (i < 4), 11 last piece + barrier B, 12 partial sums exchanged,
13 plane stores issued; 14 / 15 s_memtime at entry / end (shader clock).  Loaders: 0 entry, 1 DMAs issued, 2 stage 0 landed, 3 all."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
M = int(os.environ.get("M", 64))
shapes = {"qkv": (4608, 3584), "o": (3584, 3584), "c5_o": (2048, 4096), "down_tp8": (3584, 2432)}
for name in os.environ.get("SHAPES", "o,qkv").split(","):
    n, k = shapes[name]
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(40)]
    x = torch.randn(M, k, device=dev, dtype=torch.float16)
    ps = [Q.pack_w4a16_scales(w[1], w[2]) for w in ws]
    pw = [Q.pack_w4a16_weights(w[0]) for w in ws]
    del ws
    tl = torch.zeros(256 * 12 * 16, dtype=torch.int64, device=dev)
    os.environ.pop("LL_GEMM_SS_TIMELINE", None)
    for i in range(40):
        Q.w4a16_matmul_partials(x, pw[i], ps[i])
    torch.cuda.synchronize()
    samples = []
    for rep in range(5):
        tl.zero_()
        # a cold-ish launch: 30 other weight sets (~250 MB) went through the caches since this one was read
        for i in range(1, 31):
            Q.w4a16_matmul_partials(x, pw[i], ps[i])
        os.environ["LL_GEMM_SS_TIMELINE"] = hex(tl.data_ptr())
        Q.w4a16_matmul_partials(x, pw[0], ps[0])
        torch.cuda.synchronize()
        os.environ.pop("LL_GEMM_SS_TIMELINE", None)
        t = tl.view(256, 12, 16).cpu().double()
        live = t[:, 0, 0] > 0
        t = t[live]
        t0 = t[:, :, 0][t[:, :, 0] > 0].min()
        clk = (t[:, :8, 15] - t[:, :8, 14]) / ((t[:, :8, 13] - t[:, :8, 0]) / 100.0)
        samples.append(((t - t0) / 100.0, float(clk.median())))
    t = samples[-1][0]
    print(f"== {name} M={M}: {t.shape[0]} workgroups; launch span (first entry -> last store issued), five launches: "
          + " ".join(f"{float(s[0][:, :8, 13].max()):.2f}" for s in samples) + " us; shader clock "
          + " ".join(f"{s[1] / 1e3:.2f}" for s in samples) + " GHz")
    def row(label, waves, idx):
        v = t[:, waves, idx].reshape(-1)
        v = v[v > -1e5]
        if v.numel() == 0:
            return
        print(f"  {label:34s} median {float(v.median()):5.2f}  min {float(v.min()):5.2f}  max {float(v.max()):5.2f} us")
    C, Ld = list(range(8)), list(range(8, 12))
    row("consumer entry", C, 0); row("consumer weight loads issued", C, 1); row("consumer barrier A passed", C, 2)
    for i in range(4):
        row(f"consumer piece {i} landed", C, 3 + 2 * i); row(f"consumer piece {i} multiplied", C, 4 + 2 * i)
    row("consumer last piece + barrier B", C, 11); row("consumer sums exchanged", C, 12); row("consumer stores issued", C, 13)
    row("loader entry", Ld, 0); row("loader DMAs issued", Ld, 1); row("loader stage 0 landed", Ld, 2); row("loader all landed", Ld, 3)
    del pw, ps
    torch.cuda.empty_cache()
