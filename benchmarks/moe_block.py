"""GPU time of fused_moe at the Qwen3-30B-A3B routed block (T = 64 tokens, 128 experts, top-8, hidden 2048, expert
intermediate 768, fp8 e4m3 weights with 128 x 128 block scales), hipGraph-replayed over rotating expert stacks so the
weights come from HBM.  Prints us per block and the expert-stack bytes / time."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K

dev = "cuda"
T, E, TOPK, H, I = int(os.environ.get("T", 64)), 128, 8, 2048, 768
g = torch.Generator(device=dev).manual_seed(0)
copies = 4
w1 = [torch.randint(0, 256, (E, 2 * I, H), device=dev, dtype=torch.uint8, generator=g) & 0x77 for _ in range(copies)]
w2 = [torch.randint(0, 256, (E, H, I), device=dev, dtype=torch.uint8, generator=g) & 0x77 for _ in range(copies)]
s1 = torch.rand(E, 2 * I // 128, H // 128, device=dev) * 0.01 + 0.002
s2 = torch.rand(E, H // 128, I // 128, device=dev) * 0.01 + 0.002
x = (torch.randn(T, H, device=dev, generator=g) * 0.3).half()
logits = torch.randn(T, E, device=dev, generator=g)
tw, ti = torch.topk(torch.softmax(logits, -1), TOPK, dim=-1)
tw = (tw / tw.sum(-1, keepdim=True)).float()
f = lambda i: K.fused_moe(x, w1[i], w2[i], tw, ti, w1_scale=s1, w2_scale=s2, group_n=128, group_k=128)
f(0); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(copies):
        f(i)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    gr.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (5 * copies)
hit = len(torch.unique(ti))
byt = hit * (2 * I * H + H * I)
print(f"T={T}: {us:.1f} us per routed block, {hit}/{E} experts hit, {byt / 1e6:.0f} MB of expert weights -> {byt / us / 1e6:.2f} TB/s")
print(json.dumps({"us": round(us, 1), "experts_hit": hit, "TBps": round(byt / us / 1e6, 2)}))
