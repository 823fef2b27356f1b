"""A few eager launches of one W4A16 decode GEMM shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
from lite_llama_amd.kernels.quantization import pack_w4a16_scales
n, k = [int(v) for v in os.environ.get("SHAPE", "18944x3584").split("x")]
M = int(os.environ.get("M", 64))
dev = "cuda"
ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
       torch.rand(n, k // 128, device=dev) * 0.01 + 0.005,
       torch.randint(0, 16, (n, k // 128), device=dev).float()) for _ in range(8)]
pk = [pack_w4a16_scales(w[1], w[2]) for w in ws]
x = torch.randn(M, k, device=dev, dtype=torch.float16)
for i in range(int(os.environ.get("REPS", 16))):
    K.w4a16_matmul(x, *ws[i % 8], group_size=128, packed_scales=pk[i % 8])
torch.cuda.synchronize()
