"""In-kernel timeline of wgemm4_kernel (needs a -DV4_TIMELINE build: LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/v4_tl.so).
Stamps are s_memtime (shader clock) per (workgroup, wave); printed as medians over workgroups, in cycles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
shapes = [("gate|up", 37888, 3584, 1), ("down", 3584, 18944, 0), ("o", 3584, 3584, 0)]
if os.environ.get("ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["ONLY"].split(",")]
for name, n, k, epi in shapes:
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(3)]
    x = torch.randn(64, k, device=dev, dtype=torch.float16)
    ps = [Q.pack_w4a16_scales(w[1], w[2]) for w in ws]
    pw = [Q.pack_w4a16_weights(w[0]) for w in ws]
    run = (lambda i: Q.w4a16_matmul_prepacked(x, pw[i], ps[i], gate_up_swiglu=True)) if epi else (lambda i: Q.w4a16_matmul_partials(x, pw[i], ps[i]))
    tl = torch.zeros(512 * 12 * 64, dtype=torch.int64, device=dev)
    os.environ.pop("LL_GEMM4_TIMELINE", None)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    tl.zero_()
    os.environ["LL_GEMM4_TIMELINE"] = hex(tl.data_ptr())
    run(0)
    torch.cuda.synchronize()
    os.environ.pop("LL_GEMM4_TIMELINE", None)
    t = tl.view(512, 12, 64).cpu().double()
    live = t[:, 0, 0] > 0
    t = t[live]
    nwg = t.shape[0]
    med = lambda v: float(v.median()) if v.numel() else -1.0
    print(f"== {name}: {nwg} workgroups")
    for wv, label in [(8, "W loader 0"), (9, "W loader 1 (+scales)"), (10, "X loader 0"), (0, "consumer q0 heavy"), (4, "consumer q0 light")]:
        w = t[:, wv, :]
        e = w[:, 0:1]
        rel = lambda c: med((w[:, c] - w[:, 0])[w[:, c] > 0])
        line = f"  {label}: P0 +{rel(1):.0f}"
        if wv >= 8:
            parts = []
            for u in range(1, 12):
                a, b, c, prev = w[:, 2 + 4 * u], w[:, 3 + 4 * u], w[:, 4 + 4 * u], w[:, 4 + 4 * (u - 1)]
                m = (a > 0) & (b > 0) & (c > 0) & (prev > 0)
                if m.any():
                    parts.append("u%d %d/%d/%d" % (u, med((a - prev)[m]), med((b - a)[m]), med((c - b)[m])))
            line += " | issue/wait/barrier cycles: " + " ".join(parts)
        else:
            parts = []
            for u in range(1, 13):
                c, prev = w[:, 4 + 4 * u], w[:, 4 + 4 * (u - 1)]
                m = (c > 0) & (prev > 0)
                if m.any():
                    parts.append("%d" % med((c - prev)[m]))
            line += " | unit period cycles: " + " ".join(parts)
            line += f" | first unit barrier +{rel(4):.0f}, exchange done +{rel(60):.0f}, wave done +{rel(61):.0f}"
        print(line)
    w0 = t[:, 0, :]
    ok = (w0[:, 63] > w0[:, 62]) & (w0[:, 61] > w0[:, 0])
    if ok.any():
        clk = ((w0[ok, 61] - w0[ok, 0]) / ((w0[ok, 63] - w0[ok, 62]) / 100.0))  # cycles per us
        print(f"  shader clock over consumer wave 0's life: median {float(clk.median()) / 1e3:.3f} GHz; life {float(((w0[ok, 63] - w0[ok, 62]) / 100.0).median()):.2f} us")
    ent = t[:, 0, 0]
    print(f"  entry spread {float(ent.max() - ent.min()):.0f} cycles (per-CU clocks are not synchronised: indicative only)")
