"""Short-stream W4A16 engine (csrc/gemm_short.hip) against the unit loop (gemm_w4_v3.hip) and an fp32 reference: split-K partial
launches at the headline's q|k|v / o shapes and at TP-shard shapes.  Per shape: the engine's plan, max |sum of planes - fp32| (and
the unit loop's), us per launch of both (hipGraph replays over ~700 MB of rotating weights).  Knobs are read once per process:
LL_GEMM_SS=0 / LL_GEMM_SS_R / LL_GEMM_SS_S select the plan.  One JSON line at the end.

    python benchmarks/gemm_short.py            # M=64
    M=32 SHAPES=o,qkv python benchmarks/gemm_short.py
"""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q
from lite_llama_amd import _lib as L

dev = "cuda"
M = int(os.environ.get("M", 64))
ALL = {
    "qkv": (4608, 3584), "o": (3584, 3584), "down": (3584, 18944), "gateup": (37888, 3584),
    "qkv_tp2": (2304, 3584), "o_tp2": (3584, 1792), "down_tp8": (3584, 2432), "gateup_tp8": (4736 // 128 * 128, 3584),
    "c5_qkv": (5120, 2048), "c5_o": (2048, 4096), "l3_qkv": (6144, 4096), "l3_o": (4096, 4096),
}
names = os.environ.get("SHAPES", "qkv,o").split(",")
TIME = os.environ.get("TIME", "1") != "0"
res = {"m": M, "env": {k: v for k, v in os.environ.items() if k.startswith("LL_GEMM")}}


def plan(n, k):
    out = (ctypes.c_int32 * 8)()
    L.lib().ll_w4a16_short_plan(M, n, k, 128, out)
    return list(out)


def timed(fn, copies):
    fn(0); torch.cuda.synchronize()
    reps = max(copies, 16)
    g = torch.cuda.CUDAGraph()
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
    return best


for name in names:
    n, k = ALL[name]
    torch.manual_seed(1)
    wbytes = n * k // 2 + n * (k // 128) * 8
    copies = max(2, int(700e6 // wbytes)) if TIME else 1
    if os.environ.get("COPIES"):  # COPIES=1: the same weights every launch -- they stay in the 256-MB Infinity Cache (what a prefetch would buy)
        copies = int(os.environ["COPIES"])
    pw, ps = [], []
    ref = None
    xpad = int(os.environ.get("XPAD", 0))  # XPAD=64: activation rows 128 B further apart (row stride no multiple of 2 KB: L2 channel spread)
    x = torch.randn(M, k + xpad, device=dev, dtype=torch.float16)[:, :k]
    for c in range(copies):
        qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
        sc = torch.rand(n, k // 128, device=dev) * 0.01 + 0.005
        zr = torch.randint(0, 16, (n, k // 128), device=dev).float()
        pw.append(Q.pack_w4a16_weights(qw)); ps.append(Q.pack_w4a16_scales(sc, zr))
        if c == 0:
            nib = torch.stack([(qw >> (4 * j)) & 15 for j in range(8)], dim=-1).reshape(n, k).float()
            wd = (nib - zr.repeat_interleave(128, dim=1)) * sc.repeat_interleave(128, dim=1)
            ref = x.float() @ wd.t()
            del nib, wd
        del qw, sc, zr
    a = Q.w4a16_matmul_partials(x, pw[0], ps[0], group_size=128)
    b = Q.w4a16_matmul_partials(x, pw[0], ps[0], group_size=128, _unit_loop_engine=True)
    torch.cuda.synchronize()
    sa, sb = a.parts.sum(0), b.parts.sum(0)
    scale = ref.abs().max().item()
    ent = {"plan": plan(n, k), "planes": [a.parts.shape[0], b.parts.shape[0]],
           "err_ss": round((sa - ref).abs().max().item() / scale, 6), "err_unit": round((sb - ref).abs().max().item() / scale, 6),
           "ss_vs_unit": round((sa - sb).abs().max().item() / scale, 7)}
    if TIME:
        ent["us_ss"] = round(timed(lambda i: Q.w4a16_matmul_partials(x, pw[i], ps[i], group_size=128), copies), 2)
        ent["us_unit"] = round(timed(lambda i: Q.w4a16_matmul_partials(x, pw[i], ps[i], group_size=128, _unit_loop_engine=True), copies), 2)
    if TIME and os.environ.get("FULL") and n % 64 == 0:  # the same weights as a finished-output fused gate|up launch (row-group engine)
        ent["us_full_swiglu"] = round(timed(lambda i: Q.w4a16_matmul_prepacked(x, pw[i], ps[i], gate_up_swiglu=True), copies), 2)
    res[name] = ent
    print(name, ent, flush=True)
    del pw, ps
    torch.cuda.empty_cache()
print(json.dumps(res))
