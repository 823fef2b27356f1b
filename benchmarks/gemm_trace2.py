"""Phase timing inside the v2 W4A16 engine (debug): per-role cycle stamps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda"
buf = torch.zeros(1024 * 3 * 4, dtype=torch.int64, device=dev)
os.environ["LL_GEMM_TRACE"] = hex(buf.data_ptr())
import lite_llama_amd.kernels as K
M = int(os.environ.get("M", 64))
for n, k in [(1024, 3584), (3584, 3584), (18944, 3584), (3584, 18944)]:
    ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
           torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
           torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(3)]
    xx = torch.randn(M, k, device=dev, dtype=torch.float16)
    for it in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        K.w4a16_matmul(xx, *ws[it % 3], group_size=128)
        torch.cuda.synchronize()
    t = buf.view(-1, 3, 4).cpu().double()
    t = t[t[:, 0, 1] != 0]
    med = lambda v: float(v.median())
    units = med(t[:, 1, 3])
    print(f"N={n} K={k}: wgs={t.shape[0]} units/wg={units:.0f}")
    for r, name in enumerate(["consumer", "loader  ", "producer"]):
        extra = f" flush {med(t[:, r, 3]):.0f}" if r == 0 else ""
        print(f"   {name}: prologue {med(t[:, r, 0]):.0f}  body {med(t[:, r, 1]):.0f} (per unit {med(t[:, r, 1]) / units:.0f})  barrier-wait {med(t[:, r, 2]):.0f} (per unit {med(t[:, r, 2]) / units:.0f}){extra}")
