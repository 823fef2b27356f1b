"""Eager launches of one W4A16 shape on the pre-packed decode engine (profiling target for rocprofv3).
SHAPE=NxK (default the fused gate|up of Qwen2.5-7B), REPS launches over rotating weight copies."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

n, k = map(int, os.environ.get("SHAPE", "37888x3584").split("x"))
reps = int(os.environ.get("REPS", 20))
M = int(os.environ.get("M", 64))
dev = "cuda"
copies = 6
ws = [(torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32),
       torch.rand(n, k // 128, device=dev) * 0.01 + 0.005, torch.randint(0, 16, (n, k // 128), device=dev).float())
      for _ in range(copies)]
x = torch.randn(M, k, device=dev, dtype=torch.float16)
ps = [Q.pack_w4a16_scales(w[1], w[2]) for w in ws]
pw = [Q.pack_w4a16_weights(w[0]) for w in ws]
for i in range(reps):
    Q.w4a16_matmul_prepacked(x, pw[i % copies], ps[i % copies], group_size=128)
torch.cuda.synchronize()
print("done")
