"""M-tiled 8-bit GEMMs (csrc/gemm_w8_prefill.hip) at prefill shapes: Llama-3-8B SmoothQuant (int8 x int8, batch 32 x 512 tokens) and
the dense fp8-block projections of Qwen3-30B-A3B (batch 64 x 512): ms and T(FL)OP/s per projection, HIP events over back-to-back
launches.  LL_W8_NO_MTILED=1 in the environment times the 64-row weight-streaming tile looped over M (what ran before round 6)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lite_llama_amd.kernels.quantization as Q
from lite_llama_amd import kernels as K

dev = "cuda"
res = {"mtiled": os.environ.get("LL_W8_NO_MTILED") is None}


def timed(fn, reps=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


M = int(os.environ.get("M", 16384))
w8a8 = {}
for name, n, k in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gateup", 28672, 4096), ("down", 4096, 14336)]:
    qw = torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev)
    sc = torch.rand(n, 1, device=dev) * 0.01 + 0.005
    x = (torch.randn(M, k, device=dev) * 0.5).half()
    qa, a_s = Q.quantize_activations_int8(x)
    ms_q = timed(lambda: Q.quantize_activations_int8(x))
    ms = timed(lambda: K.smoothquant_matmul(x, qw, sc))
    w8a8[name] = {"ms_incl_quantiser": round(ms, 3), "quantiser_ms": round(ms_q, 3), "TOPs_gemm_only": round(2.0 * M * n * k / (ms - ms_q) / 1e9, 1)}
    del qw, sc, x, qa, a_s
    torch.cuda.empty_cache()
res["w8a8_llama3_8b_M%d" % M] = w8a8
res["w8a8_layer_ms"] = round(sum(v["ms_incl_quantiser"] for v in w8a8.values()), 2)

M = int(os.environ.get("M5", 32768))
fp8 = {}
for name, n, k in [("qkv", 5120, 2048), ("o", 2048, 4096)]:
    qw = torch.randint(0, 256, (n, k), dtype=torch.int64, device=dev).to(torch.uint8)
    qw[(qw & 0x7F) == 0x7F] = 0
    sc = torch.rand(n // 128, k // 128, device=dev) * 0.01 + 0.005
    x = (torch.randn(M, k, device=dev) * 0.5).half()
    ms = timed(lambda: K.w8a16_matmul(x, qw, sc, group_n=128, group_k=128))
    fp8[name] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * M * n * k / ms / 1e9, 1)}
    del qw, sc, x
    torch.cuda.empty_cache()
res["fp8_block_qwen3_30b_M%d" % M] = fp8
print(json.dumps(res))
