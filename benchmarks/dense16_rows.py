"""Round 5: the 16-bit row-group weight-streaming kernel (csrc/gemm_w16_rows.hip) against the library GEMM (F.linear, + the swiglu
launch for the fused gate|up) and the split-K 16-bit engine (dense16_linear) at the decode shapes of BASELINE config 2 and at the
lm_heads: correctness vs fp32 and us per launch (hipGraph replays over rotating weights).  One JSON line."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q
from lite_llama_amd.kernels import swiglu_forward

dev = "cuda"
res = {}
shapes = [("gateup_1.5b", 32, 17920, 1536, torch.bfloat16, True), ("lm_head_1.5b", 32, 151936, 1536, torch.bfloat16, False),
          ("down_1.5b", 32, 1536, 8960, torch.bfloat16, False), ("qkv_1.5b", 32, 2048, 1536, torch.bfloat16, False),
          ("lm_head_7b", 64, 152064, 3584, torch.float16, False), ("gateup_7b_f16", 64, 37888, 3584, torch.float16, True)]
for name, m, n, k, dt, sw in shapes:
    nbytes = n * k * 2
    copies = max(2, int(600e6 // nbytes))
    ws = [(torch.randn(n, k, device=dev) * 0.03).to(dt) for _ in range(copies)]
    x = (torch.randn(m, k, device=dev) * 0.5).to(dt)
    ref = x.float() @ ws[0].float().T
    got = Q.dense16_rows_linear(x, ws[0], gate_up_swiglu=sw)
    if sw:
        r16 = ref.to(dt).float()
        ref = (F.silu(r16[:, 0::2]) * r16[:, 1::2])
    err = (got.float() - ref).abs().max().item()
    tol = (3e-2 if dt == torch.bfloat16 else 1e-2) * ref.abs().max().item()

    def lib(i):
        y = F.linear(x, ws[i])
        return swiglu_forward(y[:, 0::2], y[:, 1::2]) if sw else y

    def rows(i):
        return Q.dense16_rows_linear(x, ws[i], gate_up_swiglu=sw)

    def old(i):
        y = Q.dense16_linear(x, ws[i])
        return swiglu_forward(y[:, 0::2], y[:, 1::2]) if sw else y

    out = {"err": round(err, 5), "tol": round(tol, 5), "ok": err <= tol}
    for label, fn in (("rows", rows), ("library", lib), ("splitk16", old)):
        try:
            fn(0); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            reps = max(copies, 8)
            with torch.cuda.graph(g):
                for i in range(reps):
                    fn(i % copies)
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * reps))
            out[label + "_us"] = round(best, 2)
            del g
        except Exception as exc:
            out[label + "_us"] = f"{type(exc).__name__}: {exc}"[:80]
    out["rows_TBps"] = round(nbytes / out["rows_us"] / 1e6, 2) if isinstance(out["rows_us"], float) else None
    res[name] = out
    print(name, out, flush=True)
    del ws, x
    torch.cuda.empty_cache()
print(json.dumps(res))
