"""add-and-normalise + projection: two launches vs one (ll_w4a16_matmul_prepacked_normed), us per pair, hipGraph replays.

    python benchmarks/norm_in_gemm.py            # the two headline sites (down -> q|k|v, o -> gate|up)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q  # noqa: E402
from lite_llama_amd.kernels.norm_act import PartialSums, skip_rmsnorm_partials  # noqa: E402
from lite_llama_amd.quantization.params import quantize_int4_groupwise  # noqa: E402

os.environ["LL_NORM_IN_GEMM"] = "1"
DEV = "cuda"
REP = int(os.environ.get("REP", 16))


def site(m, n, k, s, form, copies=6):
    g = torch.Generator().manual_seed(1)
    ws = []
    for c in range(copies):  # rotating weight copies: every launch streams from HBM
        w = torch.randn(n, k, generator=g) * 0.05
        qw, sc, zr = quantize_int4_groupwise(w.to(DEV), 128)
        ws.append((Q.pack_w4a16_weights(qw), Q.pack_w4a16_scales(sc, zr)))
    parts = (torch.randn(s, m, k, generator=g) * 0.3).to(DEV)
    res = torch.randn(m, k, generator=g).half().to(DEV)
    nw = torch.ones(k).half().to(DEV)

    def project(x, pending, i):
        pw, ps = ws[i % copies]
        if form == "partials":
            return Q.w4a16_matmul_partials(x, pw, ps, group_size=128, pending=pending)
        return Q.w4a16_matmul_prepacked(x, pw, ps, group_size=128, gate_up_swiglu=form == "swiglu", pending=pending)

    def run(fused, norm=True):
        for i in range(REP):
            if not norm:
                project(res, None, i)
                continue
            y, _ = skip_rmsnorm_partials(PartialSums(parts, (m, k), torch.float16), res, nw, 1e-6, defer=fused)
            project(y.out if fused else y, y if fused else None, i)

    out = {}
    for name, kw in (("gemm_only", dict(fused=False, norm=False)), ("two_launches", dict(fused=False)), ("one_launch", dict(fused=True))):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            run(**kw)
        torch.cuda.current_stream().wait_stream(st)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run(**kw)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            graph.replay()
        t1.record()
        torch.cuda.synchronize()
        out[name] = round(t0.elapsed_time(t1) * 1e3 / (10 * REP), 2)
    return out


if __name__ == "__main__":
    r = {"qkv(s=9)": site(64, 4608, 3584, 9, "partials"), "gateup(s=5)": site(64, 37888, 3584, 5, "swiglu", copies=3)}
    print(json.dumps(r))
