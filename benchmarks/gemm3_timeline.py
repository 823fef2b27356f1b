"""In-kernel timeline of wgemm3_kernel (needs a -DV3_TIMELINE build of gemm_w4_v3.hip; LL_LIB_OVERRIDE points at it).
Stamps are s_memrealtime (100 MHz); printed in us relative to the earliest workgroup entry."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
shapes = [("qkv", 4608, 3584, 0), ("o", 3584, 3584, 0), ("gate|up", 37888, 3584, 1), ("down", 3584, 18944, 0)]
if os.environ.get("SHAPES"):
    shapes = [(f"s{i}", *map(int, t.split("x")), 0) for i, t in enumerate(os.environ["SHAPES"].split(","))]
PARTIALS = bool(os.environ.get("PARTIALS"))  # the step's own form of the row-parallel / q|k|v launches


def gemm(x, w, s, epi):
    if PARTIALS and not epi:
        return Q.w4a16_matmul_partials(x, w, s, group_size=128)
    return Q.w4a16_matmul_prepacked(x, w, s, group_size=128, gate_up_swiglu=bool(epi))


for name, n, k, epi in shapes:
    copies = 3
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
           torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(copies)]
    x = torch.randn(64, k, device=dev, dtype=torch.float16)
    ps = [Q.pack_w4a16_scales(w[1], w[2]) for w in ws]
    pw = [Q.pack_w4a16_weights(w[0]) for w in ws]
    tl = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
    for wave in (8, 10):  # loader waves: 8 = weights + scales, 10 = activations
        os.environ["LL_GEMM3_TL_WAVE"] = str(wave)
        os.environ.pop("LL_GEMM3_TIMELINE", None)
        for i in range(copies):
            gemm(x, pw[i], ps[i], epi)
        torch.cuda.synchronize()
        tl.zero_()
        os.environ["LL_GEMM3_TIMELINE"] = hex(tl.data_ptr())
        gemm(x, pw[0], ps[0], epi)
        torch.cuda.synchronize()
        os.environ.pop("LL_GEMM3_TIMELINE", None)
        t = tl.view(1024, 64).cpu().double()
        t = t[t[:, 0] > 0]
        rel = lambda c: ((t[:, c] - t[:, 0]) / 100.0)[t[:, c] > 0]
        line = []
        for u in range(2, 10):
            a, b, c0, prev = t[:, 4 + 3 * u], t[:, 5 + 3 * u], t[:, 6 + 3 * u], t[:, 6 + 3 * (u - 1)]
            m = (a > 0) & (b > 0) & (c0 > 0) & (prev > 0)
            if m.any():
                line.append("u%d %.2f/%.2f/%.2f" % (u, ((a - prev)[m] / 100).median(), ((b - a)[m] / 100).median(), ((c0 - b)[m] / 100).median()))
        print(f"== {name} loader wave {wave}: prologue issued {rel(2).median():.2f} us, first barrier {rel(3).median():.2f}; per unit issue/wait/barrier us: " + " ".join(line))
    for wave in (0, 4):
        os.environ["LL_GEMM3_TL_WAVE"] = str(wave)
        os.environ.pop("LL_GEMM3_TIMELINE", None)
        for i in range(copies):
            gemm(x, pw[i], ps[i], epi)
        torch.cuda.synchronize()
        tl.zero_()
        os.environ["LL_GEMM3_TIMELINE"] = hex(tl.data_ptr())
        gemm(x, pw[0], ps[0], epi)
        torch.cuda.synchronize()
        os.environ.pop("LL_GEMM3_TIMELINE", None)
        t = tl.view(1024, 64).cpu().double()
        live = t[:, 0] > 0
        t = t[live]
        t0 = t[:, 0].min()

        def stat(col, rel_entry=False):
            v = t[:, col]
            m = v > 0
            if m.sum() == 0:
                return "      -      "
            base = t[m, 0] if rel_entry else t0
            d = (v[m] - base) / 100.0
            return f"{d.median():6.2f}/{d.max():6.2f}"

        print(f"== {name} wave {wave}: {int(live.sum())} workgroups; us after the first entry (median/max over workgroups)")
        print(f"  entry spread {stat(0)} | decoded {stat(1, True)} (after own entry) | loads issued {stat(2, True)} | first barrier {stat(3, True)}")
        units = [c for c in range(4, 44) if (t[:, c] > 0).any()]
        per = []
        for c in units[:12]:
            per.append(stat(c, True))
        print("  unit barriers (after own entry):", " ".join(per))
        nun = (t[:, 4:44] > 0).sum(1)
        # steady period: (last unit stamp - first unit stamp) / (n - 1)
        rows = nun > 2
        if rows.any():
            last = torch.gather(t[:, 4:44], 1, (nun.clamp(min=1) - 1).long().unsqueeze(1)).squeeze(1)
            period = ((last - t[:, 4]) / (nun - 1).clamp(min=1) / 100.0)[rows]
            print(f"  units/workgroup {int(nun.min())}..{int(nun.max())}; unit period median {period.median():.3f} us (min {period.min():.3f}, max {period.max():.3f})")
        if os.environ.get("TL_DUMP"):  # per-workgroup table: index, entry, units, first / last unit barrier, done (us after the first entry)
            idx_all = torch.nonzero(live).squeeze(1)
            with open(os.environ["TL_DUMP"] + f".{name.replace('|', '')}.w{wave}.txt", "w") as fdump:
                fdump.write("# wg entry units first_unit last_unit seg_begin done\n")
                lastu = torch.gather(t[:, 4:44], 1, (nun.clamp(min=1) - 1).long().unsqueeze(1)).squeeze(1)
                for r in range(t.shape[0]):
                    g = lambda v: (v - t0) / 100.0 if v > 0 else -1.0
                    fdump.write("%d %.2f %d %.2f %.2f %.2f %.2f\n" % (int(idx_all[r]), g(t[r, 0]), int(nun[r]), g(t[r, 4]), g(lastu[r]), g(t[r, 50]), g(t[r, 60])))
        cyc = (t[:, 62] - t[:, 61]); rt = (t[:, 60] - t[:, 0]) / 100.0
        ok = (t[:, 62] > 0) & (rt > 0)
        if ok.any():
            print(f"  shader clock over the wave's life: median {(cyc[ok] / rt[ok]).median() / 1e3:.2f} GHz")
        c = t[:, 44:49]
        m = (c > 0).all(1)
        if m.any():
            d = (c[m, 1:] - c[m, :-1])
            print("  unit 3, shader cycles (median): reads issued %d | compute issued %d | barrier %d | bookkeeping %d | step total %d" % (
                d[:, 0].median(), d[:, 1].median(), d[:, 2].median(), d[:, 3].median(), (c[m, 4] - c[m, 0]).median()))
        print(f"  last segment end: begin {stat(50)} exchanged {stat(51)} counter seen {stat(52)} merged {stat(53)} | wave done {stat(60)}")
