"""Repeatability stress of the all-of-K short-stream form (ring slot reuse, barrier protocol): every shape 300 launches back to back,
interleaved with a bandwidth hog on a second stream; each result must equal the first bit for bit and stay within 2^-9 of the unit loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
hog_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0
for (m, n, k, sw) in [(64, 3584, 3584, False), (64, 1024, 3584, False), (64, 4608, 3584, True), (64, 9472, 3584, True), (33, 4096, 4096, False),
                      (17, 1024, 4608, False), (64, 8192, 4096, True), (1, 3584, 3584, False)]:
    g = torch.Generator(device=dev).manual_seed(n + k + m)
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    sc = torch.rand(n, k // 128, device=dev, generator=g) * 0.01 + 0.005
    zr = torch.randint(0, 16, (n, k // 128), device=dev, generator=g).float()
    pw, ps = Q.pack_w4a16_weights(qw), Q.pack_w4a16_scales(sc, zr)
    x = (torch.randn(m, k, device=dev, generator=g) * 0.5).half()
    kw = dict(gate_up_swiglu=True) if sw else {}
    first = Q.w4a16_matmul_prepacked(x, pw, ps, **kw)
    unit = Q.w4a16_matmul_prepacked(x, pw, ps, _tile_blocks=1, **kw)
    ok_unit = (first.float() - unit.float()).abs().max().item() <= unit.float().abs().max().item() * 2.0 ** -9 + 1e-6
    diff = 0
    for it in range(300):
        if it % 10 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        y = Q.w4a16_matmul_prepacked(x, pw, ps, **kw)
        if not torch.equal(y, first):
            diff += 1
    torch.cuda.synchronize()
    print((m, n, k, sw), "vs unit loop ok" if ok_unit else "UNIT LOOP MISMATCH", "repeat mismatches:", diff)
    bad += diff + (0 if ok_unit else 1)
print("STRESS", "OK" if bad == 0 else "FAILED")
