"""Quick GPU probe of the streaming GEMMs under rocprofv3 (kernel times come from the trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
from oracle import oracle as O

dev = "cuda"
M = int(os.environ.get("M", 64))
torch.manual_seed(0)
# correctness spot check
x = torch.randn(M, 3584, dtype=torch.float16) * 0.5
w = torch.randn(512, 3584) * 0.05
qw, sc, zr = O.quantize_int4_groupwise(w, 128)
ref = O.w4a16_matmul(x, qw, sc, zr, group_size=128)
y = K.w4a16_matmul(x.to(dev), qw.to(dev), sc.to(dev), zr.to(dev), group_size=128)
print("max abs err", (y.float().cpu() - ref.float()).abs().max().item())
for n, k in [(3584, 3584), (1024, 3584), (18944, 3584), (3584, 18944)]:
    copies = max(2, int(600e6 // (n * k // 2)))
    ws = []
    for _ in range(copies):
        ws.append((torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
                   torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
                   torch.randint(0, 16, (n, k // 128), device=dev).float()))
    xx = torch.randn(M, k, device=dev, dtype=torch.float16)
    for it in range(20):
        q, s, z = ws[it % copies]
        K.w4a16_matmul(xx, q, s, z, group_size=128)
torch.cuda.synchronize()
