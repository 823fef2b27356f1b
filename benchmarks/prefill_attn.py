"""GPU time of flash_attention2_no_pad (varlen causal prefill) at Qwen2.5-7B head geometry; prints TFLOP/s against the
causal flop count 4 * D * S(S+1)/2 per (sequence, query head)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K

dev = "cuda"
HQ, HKV, D = int(os.environ.get("HQ", 28)), int(os.environ.get("HKV", 4)), int(os.environ.get("D", 128))
res = {}
for batch, s in ((8, 512), (4, 2048), (1, 8192), (64, 512)):
    q = torch.randn(batch * s, HQ, D, device=dev, dtype=torch.float16)
    k = torch.randn(batch * s, HKV, D, device=dev, dtype=torch.float16)
    v = torch.randn(batch * s, HKV, D, device=dev, dtype=torch.float16)
    start = (torch.arange(batch, device=dev) * s).int()
    lens = torch.full((batch,), s, device=dev, dtype=torch.int32)
    scale = 1.4426950408889634 / D ** 0.5
    for _ in range(2):
        K.flash_attention2_no_pad(q, k, v, scale, start, lens, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        K.flash_attention2_no_pad(q, k, v, scale, start, lens, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 4.0 * D * (s * (s + 1) / 2) * HQ * batch
    res[f"b{batch}xs{s}"] = {"ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1)}
    print(f"batch {batch:3d} x seq {s:5d}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
print(json.dumps(res))
