"""Randomised shape sweep of the W4A16 decode engine against the generic engine (LL_GEMM_V1=1) and,
for small shapes, the CPU oracle: exercises the stream-K / tile-group plans' edge cases."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels as K
from lite_llama_amd.kernels.quantization import pack_w4a16_scales
random.seed(int(os.environ.get("SEED", 0)))
dev = "cuda"
bad = 0
for it in range(int(os.environ.get("CASES", 80))):
    m = random.choice([1, 2, 7, 16, 31, 32, 33, 48, 63, 64])
    n = 128 * random.choice([1, 2, 3, 5, 8, 9, 17, 28, 36, 61, 148, 255, 256, 257, 300])
    k = 128 * random.choice([1, 2, 3, 4, 5, 7, 8, 13, 28, 29, 37, 64, 148])
    gs = random.choice([128, 128, 256]) if k % 256 == 0 else 128
    g = torch.Generator(device=dev).manual_seed(it)
    x = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, generator=g, device=dev).to(torch.int32)
    sc = torch.rand(n, k // gs, generator=g, device=dev) * 0.01 + 0.005
    zr = torch.randint(0, 16, (n, k // gs), generator=g, device=dev).float()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half() if it % 3 == 0 else None
    os.environ.pop("LL_GEMM_V1", None)
    y2 = K.w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias, packed_scales=pack_w4a16_scales(sc, zr) if it % 2 else None)
    y2b = K.w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias)
    os.environ["LL_GEMM_V1"] = "1"
    y1 = K.w4a16_matmul(x, qw, sc, zr, group_size=gs, bias=bias)
    torch.cuda.synchronize()
    err = (y2.float() - y1.float()).abs().max().item()
    scale = y1.float().abs().max().item() + 1e-6
    ok = err <= 2e-3 * scale + 2e-3 and torch.equal(y2, y2b)
    if not ok:
        bad += 1
        print(f"MISMATCH m={m} n={n} k={k} gs={gs}: max err {err:.4g} (scale {scale:.3g}) packed==unpacked {torch.equal(y2, y2b)}")
print(f"{bad} mismatches")
