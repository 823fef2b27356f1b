"""Eager flash_decoding launches at the headline decode shape (profiling target)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
dev = "cuda"
B, HQ, HKV, D, ctx = 64, 28, 4, 128, int(os.environ.get("CTX", 512))
rows = B * ctx
pools = [torch.randn(rows, 2 * HKV, D, device=dev, dtype=torch.float16) * 0.5 for _ in range(6)]
q = torch.randn(B, HQ, D, device=dev, dtype=torch.float16) * 0.3
table = torch.arange(rows, device=dev, dtype=torch.int32).view(B, ctx)
req = torch.arange(B, dtype=torch.int32, device=dev)
seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
for i in range(18):
    K.flash_decoding(q, pools[i % 6][:, :HKV], pools[i % 6][:, HKV:], 1.0 / D**0.5, table, req, seq, ctx)
torch.cuda.synchronize()
print("done")
