"""Phase timing inside the stream-K GEMM (debug): per-workgroup cycle stamps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda"
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["LL_GEMM_TRACE"] = hex(buf.data_ptr())
import lite_llama_amd.kernels as K
M = int(os.environ.get("M", 64))
for n, k in [(512, 3584), (1024, 3584), (3584, 3584), (18944, 3584), (3584, 18944)]:
    ws = []
    for _ in range(3):
        ws.append((torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
                   torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
                   torch.randint(0, 16, (n, k // 128), device=dev).float()))
    xx = torch.randn(M, k, device=dev, dtype=torch.float16)
    for it in range(3):
        buf.zero_()
        q, s, z = ws[it % 3]
        torch.cuda.synchronize()
        K.w4a16_matmul(xx, q, s, z, group_size=128)
        torch.cuda.synchronize()
    t = buf.view(-1, 8).cpu()
    t = t[t[:, 0] != 0].double()
    units = t[:, 4]
    med = lambda v: float(v.median())
    rs = (t[:, 5] - t[:, 5].min()) / 100.0   # us (100 MHz)
    re = (t[:, 6] - t[:, 5].min()) / 100.0
    import numpy as np
    dur = (re - rs)
    order = dur.argsort(descending=True)[:5]
    for i in order.tolist():
        print(f"   slow wg: dur {dur[i]:.1f} us  start {rs[i]:.1f}  prologue {t[i,1]-t[i,0]:.0f}  loop {t[i,2]-t[i,1]:.0f}  flush {t[i,3]-t[i,2]:.0f} cwait {t[i,7]:.0f} units {t[i,4]:.0f}")
    print(f"   dur p50 {dur.median():.1f} p90 {dur.kthvalue(int(0.9*len(dur)))[0]:.1f} max {dur.max():.1f}")
    print("   start-time histogram (us):", np.histogram(rs.numpy(), bins=8)[0].tolist(), f"first..last start {rs.min():.1f}..{rs.max():.1f}  last end {re.max():.1f}  median dur {med(re-rs):.1f}")
    print(f"N={n} K={k}: wgs={t.shape[0]} units/wg={units.mean():.1f} | prologue {med(t[:,1]-t[:,0]):.0f} | loop {med(t[:,2]-t[:,1]):.0f} "
          f"(per unit {med((t[:,2]-t[:,1])/units):.0f}) | flush {med(t[:,3]-t[:,2]):.0f} | consumer barrier-wait/unit {med(t[:,7]/units):.0f} "
          f"| producer work/unit {med(t[:,5]/units):.0f} barrier-wait/unit {med(t[:,6]/units):.0f}")
