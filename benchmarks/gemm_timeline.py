"""In-kernel timeline of the W4A16 decode GEMM (debug build: LL_EXTRA_HIPCC_FLAGS=-DV2_TIMELINE).

Per role (consumer wave 0, loader wave 8, producer wave 10) and unit: when the role had its data
ready, when it arrived at the unit barrier and when it left; s_memrealtime ticks (10 ns).
 """
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda"
TLU, TLN = 40, 6 + 3 * 40
DONE, PA, PB, PC = 2 + 3 * 40, 3 + 3 * 40, 4 + 3 * 40, 5 + 3 * 40
buf = torch.zeros(512 * 3 * TLN, dtype=torch.int64, device=dev)
os.environ["LL_GEMM_TIMELINE"] = hex(buf.data_ptr())
import lite_llama_amd.kernels as K
from lite_llama_amd.kernels.quantization import pack_w4a16_scales
M = int(os.environ.get("M", 64))
TICK = 0.01  # us
for n, k in [(3584, 3584), (37888, 3584), (3584, 18944)]:
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
           torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(6)]
    pk = [pack_w4a16_scales(w[1], w[2]) for w in ws]
    x = torch.randn(M, k, device=dev, dtype=torch.float16)
    for it in range(6):
        buf.zero_()
        torch.cuda.synchronize()
        K.w4a16_matmul(x, *ws[it], group_size=128, packed_scales=pk[it])
        torch.cuda.synchronize()
    t = buf.view(512, 3, TLN).cpu().double()
    live = t[:, 0, 0] != 0
    t = t[live]
    wgs = t.shape[0]
    t0 = t[:, :, 0].min()
    rel = lambda a: (a - t0) * TICK
    print(f"\nN={n} K={k} M={M}: {wgs} workgroups; kernel span {rel(t[:, 0, DONE].max()):.2f} us (first entry -> last consumer done)")
    print(f"  entry spread {rel(t[:, 0, 0].max()):.2f} us; prologue barrier passed at {rel(t[:, 0, 1].median()):.2f} us (median), last {rel(t[:, 0, 1].max()):.2f}")
    med = lambda a: float(((a - t[:, :1, 0].min(1).values) * TICK).median()) if a.dim() == 1 else 0.0
    ent = t[:, 0, 0]
    print(f"  prologue, median us after the workgroup's own entry: table built {((t[:, 0, PA] - ent) * TICK).median():.2f} | "
          f"loader: loads issued {((t[:, 1, PB] - ent) * TICK).median():.2f}, units 0-1 stored {((t[:, 1, PC] - ent) * TICK).median():.2f} | "
          f"producer: loads issued {((t[:, 2, PB] - ent) * TICK).median():.2f}, x tiles 0-1 stored {((t[:, 2, PC] - ent) * TICK).median():.2f} | "
          f"prologue barrier passed {((t[:, 0, 1] - ent) * TICK).median():.2f}")
    units = int(((t[:, 0, 4:2 + 3 * TLU:3] != 0).sum(1)).median())
    print(f"  units per workgroup (median, capped at {TLU}): {units}")
    names = ["consumer", "loader  ", "producer"]
    # per-unit: arrival at the barrier per role (relative to the barrier release = max arrival ~ leave time)
    for v in list(range(min(units, 12))) + ([units - 1] if units > 12 else []):
        leave = t[:, 0, 2 + 3 * v + 2]
        ok = leave != 0
        if ok.sum() == 0:
            continue
        row = []
        for r in range(3):
            arr = t[ok, r, 2 + 3 * v + 1]
            row.append(f"{names[r]} waits {((leave[ok] - arr) * TICK).median():5.2f}")
        ready = (t[ok, 1, 2 + 3 * v] - t[ok, 1, 2 + 3 * (v - 1) + 2]) * TICK if v > 0 else None
        per = (leave[ok] - (t[ok, 0, 2 + 3 * (v - 1) + 2] if v > 0 else t[ok, 0, 1])) * TICK
        extra = f"  loader data-wait+store {ready.median():5.2f}" if ready is not None else ""
        print(f"  unit {v:3d}: period {per.median():5.2f} us | " + " | ".join(row) + extra)
    done = rel(t[:, 0, DONE])
    lastleave = rel(torch.stack([t[i, 0, 2 + 3 * (min(int((t[i, 0, 4:2 + 3 * TLU:3] != 0).sum()), TLU) - 1) + 2] for i in range(wgs)]))
    print(f"  last unit barrier left at {lastleave.median():.2f} us (median), max {lastleave.max():.2f}; consumer done at {done.median():.2f} (median), max {done.max():.2f}")
    tail = (done - lastleave)
    print(f"  tail after the last unit (k-reduction + flush/merge): median {tail.median():.2f} us, max {tail.max():.2f} us")
