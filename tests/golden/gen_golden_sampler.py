"""Golden vectors for the sampler row from the REFERENCE sampler (build container only).

    python tests/golden/gen_golden_sampler.py

Imports lite_llama.engine.sampler from /root/reference (pure torch, CPU).  Deterministic pieces only:
``apply_repetition_penalty`` outputs, and for nucleus sampling the filtered + renormalised
distribution that ``sample_top_p`` hands to ``torch.multinomial`` (captured by wrapping
``torch.multinomial``; the random draw itself cannot be pinned).  Saved: plain arrays.
"""

import os
import sys

sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from lite_llama.engine import sampler as S

    g = torch.Generator().manual_seed(21)
    out = {}
    # ---- repetition penalty: scalar and per-row factors, padded spans with duplicate ids ----
    for name, dtype in (("f32", torch.float32), ("f16", torch.float16)):
        b, v, span = 5, 300, 12
        logits = (torch.randn(b, v, generator=g) * 3).to(dtype)
        ids = torch.randint(0, v, (b, span), generator=g)
        ids[:, 3] = ids[:, 1]                       # duplicates: penalty must stay idempotent
        mask = torch.rand(b, span, generator=g) > 0.3
        ids[0, 5] = ids[0, 6]
        mask[0, 5], mask[0, 6] = True, False        # same id once real, once padded
        gen = S.GeneratedSpan(token_ids=ids, mask=mask)
        out[f"rp_{name}.logits"] = logits.float().numpy()
        out[f"rp_{name}.ids"] = ids.numpy()
        out[f"rp_{name}.mask"] = mask.numpy()
        out[f"rp_{name}.scalar_1p3"] = S.apply_repetition_penalty(logits, gen, 1.3).float().numpy()
        pen = torch.tensor([1.0, 1.1, 1.5, 0.8, 2.0]).view(b, 1)
        out[f"rp_{name}.row_penalty"] = pen.numpy()
        # per-row factors are float32 [batch, 1] tensors in the reference (BatchedSamplingParams.build):
        # with fp16 logits the result is promoted to float32
        out[f"rp_{name}.per_row"] = S.apply_repetition_penalty(logits, gen, pen).float().numpy()

    # ---- nucleus filter: capture what sample_top_p passes to multinomial ----
    b, v = 6, 997
    logits = torch.randn(b, v, generator=g) * 2.5
    temperature = torch.tensor([1.0, 0.6, 0.3, 1.5, 0.8, 1.0]).view(b, 1)
    top_p = torch.tensor([0.9, 0.5, 0.95, 0.3, 1.0, 0.01]).view(b, 1)
    probs = torch.softmax(logits / temperature, dim=-1)
    captured = {}
    real_multinomial = torch.multinomial

    def spy(p, num_samples, *a, **k):
        captured["p"] = p.clone()
        return real_multinomial(p, num_samples, *a, **k)

    torch.multinomial = spy
    try:
        # the reference sorts in place on a copy: rebuild the index order of the filtered distribution
        sorted_probs, sorted_idx = torch.sort(probs, dim=-1, descending=True)
        S.sample_top_p(probs.clone(), top_p)
    finally:
        torch.multinomial = real_multinomial
    dist = torch.zeros_like(probs)
    dist.scatter_(1, sorted_idx, captured["p"])
    out["topp.logits"] = logits.numpy()
    out["topp.temperature"] = temperature.view(-1).numpy()
    out["topp.top_p"] = top_p.view(-1).numpy()
    out["topp.dist"] = dist.numpy()                  # renormalised nucleus distribution, token order
    np.savez_compressed(os.path.join(HERE, "sampler_reference.npz"), **out)
    print("wrote sampler_reference.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
