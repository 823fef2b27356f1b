"""Sampling step at decode shape (batch 64 x vocab 152064, fp16 logits): the HIP kernels against the
reference-shaped torch sequence (softmax + full sort + cumsum + multinomial; scatter + selects)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_llama_amd.sampling import apply_repetition_penalty, sample_top_p
dev = "cuda"
B, V = 64, 152064
logits = (torch.randn(B, V, device=dev) * 3).half()
ids = torch.randint(0, V, (B, 256), device=dev)
mask = torch.ones(B, 256, dtype=torch.bool, device=dev)
t = torch.full((B,), 0.6, device=dev); p = torch.full((B,), 0.9, device=dev); u = torch.rand(B, device=dev)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def torch_top_p():
    probs = torch.softmax(logits / 0.6, dim=-1)
    sp, si = torch.sort(probs, dim=-1, descending=True)
    cum = torch.cumsum(sp, dim=-1)
    sp[cum - sp > 0.9] = 0.0
    sp.div_(sp.sum(dim=-1, keepdim=True))
    return torch.gather(si, -1, torch.multinomial(sp.float(), 1))

def torch_penalty():
    seen = torch.zeros(B, V + 1, dtype=torch.bool, device=dev)
    seen.scatter_(1, torch.where(mask, ids, V), True)
    pen = torch.where(logits < 0, logits * 1.1, logits / 1.1)
    return torch.where(seen[:, :V], pen, logits)

print(f"top-p sampling : HIP {timeit(lambda: sample_top_p(logits, t, p, uniform=u)):8.1f} us   torch sequence {timeit(torch_top_p):8.1f} us")
print(f"repetition pen.: HIP {timeit(lambda: apply_repetition_penalty(logits, ids, mask, 1.1)):8.1f} us   torch sequence {timeit(torch_penalty):8.1f} us")
