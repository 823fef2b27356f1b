"""dense16 (hand-written 16-bit weight-streaming GEMM) vs torch F.linear (hipBLASLt) at decode shapes: the Qwen2.5-1.5B
linears at batch 32 (BASELINE config 2), the Qwen2.5-7B lm_head and the Qwen3-30B-A3B router at batch 64.  hipGraph replays over
rotating weight copies; us per launch and TB/s of weight bytes."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from lite_llama_amd.kernels.quantization import dense16_linear
dev = "cuda"
shapes = [("1.5b q", 32, 1536, 1536), ("1.5b kv", 32, 512, 1536), ("1.5b qkv", 32, 2048, 1536), ("1.5b o", 32, 1536, 1536),
          ("1.5b gate|up", 32, 17920, 1536), ("1.5b down", 32, 1536, 8960), ("1.5b lm_head", 32, 151936, 1536),
          ("7b lm_head", 64, 152064, 3584), ("moe router", 64, 128, 2048)]
res = {}
for dt in (torch.float16, torch.bfloat16):
    for name, m, n, k in shapes:
        wb = n * k * 2
        copies = max(2, min(32, int(600e6 // wb)))
        ws = [(torch.randn(n, k, device=dev) * 0.02).to(dt) for _ in range(copies)]
        x = (torch.randn(m, k, device=dev) * 0.5).to(dt)
        ref = F.linear(x, ws[0]).float(); got = dense16_linear(x, ws[0])
        err = (got.float() - ref).abs().max().item() if got is not None else float("nan")
        row = {}
        for label, fn in (("dense16", lambda i: dense16_linear(x, ws[i])), ("library", lambda i: F.linear(x, ws[i]))):
            fn(0); torch.cuda.synchronize()
            reps = max(copies, 16)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(reps):
                    fn(i % copies)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            row[label] = e0.elapsed_time(e1) * 1e3 / (4 * reps)
        res[f"{name} {str(dt)[6:]}"] = row
        print(f"{name:14s} {str(dt)[6:]:9s} M={m:2d} N={n:6d} K={k:5d}: dense16 {row['dense16']:8.2f} us ({wb / row['dense16'] / 1e6:5.2f} TB/s)   library {row['library']:8.2f} us ({wb / row['library'] / 1e6:5.2f} TB/s)   max|diff| {err:.4f}", flush=True)
        del ws
        torch.cuda.empty_cache()
print(json.dumps(res))
