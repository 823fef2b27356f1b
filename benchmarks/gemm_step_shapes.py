"""The four W4A16 GEMM launches of one Qwen2.5-7B decoder layer (fused qkv, o, fused gate/up, down) at
batch 64, a few eager launches each -- the workload for the rocprofv3 --pmc FETCH_SIZE pass that
backs bench.py's roofline.traffic (profiles/pmc_traffic.json)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
from lite_llama_amd.kernels.quantization import pack_w4a16_scales
M = int(os.environ.get("M", 64))
dev = "cuda"
shapes = [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944)]
data = []
for n, k in shapes:
    w = (torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
         torch.rand(n, k // 128, device=dev) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=dev).float())
    data.append((torch.randn(M, k, device=dev, dtype=torch.float16), w, pack_w4a16_scales(w[1], w[2])))
for _ in range(int(os.environ.get("REPS", 8))):
    for x, w, pk in data:
        K.w4a16_matmul(x, *w, group_size=128, packed_scales=pk)
torch.cuda.synchronize()
