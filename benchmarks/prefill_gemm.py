"""M-tiled W4A16 GEMM (csrc/gemm_w4_prefill.hip) at the prefill shapes of Qwen2.5-7B (batch 64 x 512 tokens): ms and TFLOP/s per
projection, HIP events over back-to-back launches.  LL_LIB_OVERRIDE selects an A/B build."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q

dev = "cuda"
M = int(os.environ.get("M", 32768))
res = {"lib": os.environ.get("LL_LIB_OVERRIDE", "default")}
for name, n, k, sw in [("qkv", 4608, 3584, False), ("o", 3584, 3584, False), ("gateup", 37888, 3584, True), ("down", 3584, 18944, False)]:
    qw = torch.randint(-(2**31), 2**31 - 1, (n, k // 8), dtype=torch.int64, device=dev).to(torch.int32)
    sc = torch.rand(n, k // 128, device=dev) * 0.01 + 0.005
    zr = torch.randint(0, 16, (n, k // 128), device=dev).float()
    pw, ps = Q.pack_w4a16_weights(qw), Q.pack_w4a16_scales(sc, zr)
    x = (torch.randn(M, k, device=dev) * 0.5).half()
    for _ in range(2):
        y = Q.w4a16_matmul_prepacked_rows(x, pw, ps, gate_up_swiglu=sw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        y = Q.w4a16_matmul_prepacked_rows(x, pw, ps, gate_up_swiglu=sw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    res[name] = [round(ms, 3), round(2.0 * M * n * k / ms / 1e9, 1)]
    del qw, sc, zr, pw, ps, x, y
    torch.cuda.empty_cache()
res["layer_ms"] = round(sum(v[0] for k_, v in res.items() if isinstance(v, list)), 2)
print(json.dumps(res))
