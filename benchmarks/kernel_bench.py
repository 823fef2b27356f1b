"""Per-kernel microbenchmarks at the shapes of BASELINE.json's configs (GPU box only).

    python benchmarks/kernel_bench.py [--iters 50]

Prints one line per kernel: average time (HIP events on the launch stream) and the
achieved fraction of the 8 TB/s HBM peak computed from ALGORITHMIC bytes.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import lite_llama_amd.kernels as K  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, secs, nbytes, extra=""):
    print(json.dumps({"kernel": name, "us": round(secs * 1e6, 2), "GBps": round(nbytes / secs / 1e9, 1),
                      "hbm_frac": round(nbytes / secs / PEAK, 4), "note": extra}), flush=True)


def rand_int4(n, k, g, dev):
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
    sc = torch.rand(n, k // g, device=dev) * 0.01 + 0.005
    zr = torch.randint(0, 16, (n, k // g), device=dev).float()
    return qw, sc, zr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--m", type=int, default=64)
    args = ap.parse_args()
    dev = "cuda"
    M = args.m
    # Enough distinct weight copies that the 256 MiB Infinity Cache cannot hold the stream.
    for name, n, k in [("w4a16 q/o 3584x3584", 3584, 3584), ("w4a16 kv 1024x3584", 1024, 3584),
                       ("w4a16 gate/up 18944x3584", 18944, 3584), ("w4a16 down 3584x18944", 3584, 18944)]:
        wbytes = n * k // 2 + 2 * n * (k // 128) * 4
        copies = max(2, int(600e6 // wbytes))
        ws = [rand_int4(n, k, 128, dev) for _ in range(copies)]
        x = torch.randn(M, k, device=dev, dtype=torch.float16)
        i = [0]

        def fn():
            qw, sc, zr = ws[i[0] % copies]
            i[0] += 1
            K.w4a16_matmul(x, qw, sc, zr, group_size=128)

        report(name + f" M={M}", timeit(fn, args.iters), wbytes, f"{copies} rotating weight copies")

    # 8-bit
    for name, n, k, fmt in [("w8a16 fp8 4096x2048", 4096, 2048, "fp8"), ("w8a8 14336x4096", 14336, 4096, "i8i8"),
                            ("w8a16 int8 14336x4096", 14336, 4096, "i8")]:
        copies = max(2, int(600e6 // (n * k)))
        if fmt == "fp8":
            ws = [(torch.randint(0, 120, (n, k), device=dev, dtype=torch.uint8),
                   torch.rand(n // 128, k // 128, device=dev) + 0.5) for _ in range(copies)]
        else:
            ws = [(torch.randint(-127, 127, (n, k), device=dev, dtype=torch.int8),
                   torch.rand(n, 1, device=dev) * 0.01) for _ in range(copies)]
        x = torch.randn(M, k, device=dev, dtype=torch.float16)
        i = [0]

        def fn():
            qw, sc = ws[i[0] % copies]
            i[0] += 1
            if fmt == "fp8":
                K.w8a16_matmul(x, qw, sc, group_n=128, group_k=128)
            elif fmt == "i8":
                K.w8a16_matmul(x, qw, sc, group_n=1, group_k=k)
            else:
                K.smoothquant_matmul(x, qw, sc)

        report(name + f" M={M}", timeit(fn, args.iters), n * k, f"{copies} rotating weight copies")

    # flash decoding: Qwen2.5-7B geometry, B=64
    for ctx in (576, 2048):
        B, hq, hkv, d = 64, 28, 4, 128
        layers = max(2, int(600e6 // (B * ctx * 2 * hkv * d * 2)))
        pools = [torch.randn(B * ctx, 2 * hkv, d, device=dev, dtype=torch.float16) for _ in range(layers)]
        table = torch.arange(B * ctx, device=dev, dtype=torch.int32).view(B, ctx)
        req = torch.arange(B, device=dev, dtype=torch.int32)
        seq = torch.full((B,), ctx, device=dev, dtype=torch.int32)
        q = torch.randn(B, hq, d, device=dev, dtype=torch.float16)
        i = [0]

        def fn():
            p = pools[i[0] % layers]
            i[0] += 1
            K.flash_decoding(q, p[:, :hkv], p[:, hkv:], 1 / math.sqrt(d), table, req, seq, ctx)

        report(f"flash_decoding B64 ctx{ctx}", timeit(fn, args.iters), B * ctx * 2 * hkv * d * 2,
               f"{layers} rotating pools")

    # small row kernels
    x = torch.randn(64, 3584, device=dev, dtype=torch.float16)
    r = torch.randn(64, 3584, device=dev, dtype=torch.float16)
    w = torch.randn(3584, device=dev, dtype=torch.float16)
    report("skip_rmsnorm 64x3584", timeit(lambda: K.skip_rmsnorm(x, r, w, 1e-6), args.iters), 64 * 3584 * 2 * 5)
    a = torch.randn(64, 18944, device=dev, dtype=torch.float16)
    b = torch.randn(64, 18944, device=dev, dtype=torch.float16)
    report("swiglu 64x18944", timeit(lambda: K.swiglu_forward(a, b), args.iters), 64 * 18944 * 2 * 3)


if __name__ == "__main__":
    main()
