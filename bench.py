"""Headline benchmark: decode throughput of Qwen2.5-7B W4A16 (group 128) at batch 64 on MI355X.

    python bench.py [--gpus N --steps K --warmup W]        # N > 1: starts its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # the same, ranks started by the caller

A "step" is ONE decode step of the whole batch (one new token per sequence) through the HIP hot
path, replayed from a hipGraph: embedding -> 28 x [skip_rmsnorm, w4a16 q/kv, rope, KV scatter,
flash_decoding, w4a16 o (+TP all-reduce), skip_rmsnorm, w4a16 gate/up, swiglu, w4a16 down
(+all-reduce)] -> norm -> fp16 lm_head -> greedy argmax -> on-device metadata advance.
Synthetic data (no checkpoints offline): seeded random fp16 weights quantised with the reference's
int4 quantiser semantics, seeded random K/V for a 512-token context per sequence, random first
tokens; greedy, no EOS stop, lockstep batch -- the reference's measurement protocol
(benchmarks/common.py:101-137).  Inputs are resident in HBM when the timed region starts.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (wgemm3_kernel = W4A16 dequant-GEMM): algorithmic weight bytes per
                  launch / average launch duration measured live with HIP events, vs 8 TB/s HBM peak
  cpu_baseline -- the reference's CPU-runnable case (benchmarks/bench_hf_baseline.py protocol: Qwen2.5-0.5B
                  geometry, batch 1, 2 x 8-token warm-up, TTFT, TPOT, 256 generated tokens) on the host cores,
                  plus "oracle_port": the CPU oracle on a slice of the headline workload (N = 1 only)
  graph        -- true when the timed steps were hipGraph replays; a refused capture is reported here (and on
                  stderr) together with "graph_error", never silently
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM = 8.0e12  # B/s, MI355X HBM3E peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults = the reference's protocol (benchmarks/common.py:101-137): 128 generated tokens after a 512-token
    # prompt, two 8-token warm-ups
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="qwen2.5-7b")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=512, help="context tokens per sequence when decoding starts")
    ap.add_argument("--quant", default="int4", choices=["int4", "int8", "smoothquant", "fp8", "none"])
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"],
                    help="activation / unquantised-weight dtype; bf16 (BASELINE config 2) needs --quant none: the reference's "
                         "quantised GEMMs are fp16-only (kernels/quantization/*.py raise on anything else)")
    ap.add_argument("--steps-per-graph", type=int, default=1,
                    help="decode steps captured in ONE hipGraph launch (the timed K steps are then K / this many replays)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--scattered", action="store_true", help="context rows in random pool order (gather cost)")
    ap.add_argument("--kv-block-size", type=int, default=0, help="block-granular KV paging on the device (0: the reference's bump allocator)")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "rccl", "oneshot"],
                    help="TP collective: RCCL, the one-shot peer-to-peer kernel, or (auto) the kernel if it passes a start-up "
                         "check against RCCL on this hardware")
    ap.add_argument("--reference-order", action="store_true",
                    help="also time the DROP-IN route: the reference's own 12-launch decoder layer through the 16 kernel names only "
                         "(what integration.install() gives a reference caller), as a second labelled value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second measured point (SURVEY 8d: prompt 2048) that the N = 1 line carries as \"secondary\"")
    ap.add_argument("--no-prefill", action="store_true", help="skip the timed prefill (TTFT) object of the N = 1 line")
    ap.add_argument("--as-secondary", action="store_true",
                    help="(used by the headline run for its \"secondary\" entries) one bounded line for another BASELINE.json "
                         "configuration: timed steps + roofline + the oracle-layer parity check, no host-side baselines")
    ap.add_argument("--no-secondary-configs", action="store_true",
                    help="keep the ctx-2048 point but skip BASELINE.json configs 2 / 4 / 5 in \"secondary\"")
    ap.add_argument("--keep-reference-weights", action="store_true",
                    help="keep the reference-format int4 tensors next to the decode engine's layout (2 x the int4 payload)")
    ap.add_argument("--cpu-layers", type=int, default=1, help="decoder layers in the CPU oracle sample")
    ap.add_argument("--shard-sim", default=None,
                    help="comma list of TP degrees (default for the N = 1 headline line: 2,4,8; 'none' to skip): per degree ONE rank's "
                         "shard shapes on this GPU with every all-reduce replaced by a same-size local copy -- one rank's compute, no "
                         "collective: a ceiling of the scaling curve, not a scaling point (\"shard_sim\" in the line)")
    ap.add_argument("--as-shard-sim", type=int, default=0,
                    help="(used by --shard-sim) run as rank 0 of this TP degree with simulated collectives; bounded line")
    ap.add_argument("--parallelism", default="tp", choices=["tp", "dp"],
                    help="--gpus N > 1: tp (default; BASELINE.json's metric is the TP = 1/2/4/8 curve) = the largest tensor-parallel "
                         "degree with a shard plan, remaining ranks as replicas; dp = N independent replicas of the one-GPU step "
                         "(no collective on the data path, weak scaling: batch 64 PER replica)")
    ap.add_argument("--as-pmc-probe", action="store_true",
                    help="(used by the live counter leg) build a 3-layer model of --model / --quant at its real widths, issue the "
                         "dense projections' launch list eagerly a few times and exit: the workload of a rocprofv3 --pmc pass")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the live rocprofv3 --pmc passes (roofline.traffic / mfma_util then carry the stored profile values)")
    ap.add_argument("--no-reference-order", action="store_true",
                    help="skip the bounded drop-in-route entry (\"reference_order\") of the default N = 1 int4 line")
    return ap.parse_args()


def algorithmic_bytes(geo, quant, batch, ctx, tp):
    """weight bytes (reference storage format) + fp16 lm_head + K/V bytes, per step, per rank."""
    per_w = {"int4": 0.5 + 8.0 / 128, "int8": 1.0, "smoothquant": 1.0, "fp8": 1.0, "none": 2.0}[quant]
    attn = geo.hidden_size * geo.q_size * 2 + geo.hidden_size * 2 * geo.kv_size
    if geo.num_experts:
        # routed experts: only the experts hit by the batch are read (in expectation, uniform routing:
        # n_e = E (1 - (1 - 1/E)^(top_k * batch))), plus the fp16 router -- SURVEY 8(d) config 5
        e, k = geo.num_experts, geo.num_experts_per_tok
        n_e = e * (1.0 - (1.0 - 1.0 / e) ** (k * batch))
        mlp = n_e * 3 * geo.hidden_size * geo.moe_intermediate_size
        router = e * geo.hidden_size * 2 / per_w  # fp16, replicated (divided back below)
        lin = geo.num_layers * (attn + mlp + router)
    else:
        lin = geo.num_layers * (attn + 3 * geo.hidden_size * geo.intermediate_size)
    w_lin = lin * per_w / tp
    w_head = geo.vocab_size * geo.hidden_size * 2
    kv_tok = geo.num_layers * 2 * geo.kv_size * 2 / tp
    return w_lin, w_head, batch * ctx * kv_tok


def projection_launches(model, quant):
    """The launch list of the dense projections as the decode step issues them: (callable, input width, weights in the launch) --
    fused [q|k|v] left as split-K partials, fused [gate|up] + swiglu, row-parallel projections in split-K partial mode at TP = 1."""
    from lite_llama_amd.distributed.parallel_state import collective_forced, get_tp_world_size
    from lite_llama_amd.linear import LinearBase, MergedColumnLinear, RowParallelLinear

    # rank 0 times these launches ALONE: nothing in the list may communicate.  At TP = 1 a row-parallel projection runs as
    # in the step (split-K partial mode); under TP the step's launch is the plain GEMM followed by the all-reduce -- only
    # the GEMM is timed here (calling the layer itself would issue a collective the other ranks never join)
    solo = get_tp_world_size() == 1 and not collective_forced()
    merged = [m for mod in model.modules() for m in vars(mod).values() if isinstance(m, MergedColumnLinear)]
    fused_members = set()
    launches_list = []  # (callable, input_size, weights in the launch)
    for mc in merged:
        if (mc.layers[0].quant is not None or quant == "none") and mc.refresh():
            fused_members.update(id(l) for l in mc.layers)
            if mc.interleave:
                fn = lambda x, mc=mc: mc.swiglu(x)
            else:  # the decode step leaves the fused q|k|v projection as split-K partials for the attention kernel
                fn = lambda x, mc=mc: (mc.partials(x) or mc(x))
            launches_list.append((fn, mc.layers[0].input_size, sum(l.input_size * l.output_size for l in mc.layers)))
    for m in model.modules():
        if isinstance(m, LinearBase) and (m.quant is not None or quant == "none") and id(m) not in fused_members:
            fn = (lambda x, m=m: m(x, partials_ok=True)) if (isinstance(m, RowParallelLinear) and solo) else m.apply_linear
            launches_list.append((fn, m.input_size, m.input_size * m.output_size))
    return launches_list, solo


def gemm_roofline(model, batch, quant, iters=6, live_pmc=None, time_it=True):
    """Average launch duration of the dominant kernel (the weight-streaming dequant-GEMM) over every
    projection of the model with its real weights -- the same calls the decode step makes (fused [q|k|v],
    fused [gate|up] + swiglu, row-parallel projections in split-K partial mode) -- by HIP events on the
    launch stream around hipGraph replays of the launch list (eager back-to-back launches as fallback).
    ``live_pmc``: the result of :func:`live_pmc_counters` of THIS run (HBM-side traffic + MFMA utilisation from rocprofv3 --pmc
    passes over the same launch list); without it the stored profile values are reported, labelled so."""
    launches_list, solo = projection_launches(model, quant)
    if not launches_list:
        return None
    dev = next(model.parameters()).device
    adt = next(model.parameters()).dtype if quant == "none" else torch.float16
    xs = {k: (torch.randn(batch, k, device=dev) * 0.5).to(adt) for _, k, _ in launches_list}
    per_w = {"int4": 0.5 + 8.0 / 128, "int8": 1.0, "smoothquant": 1.0, "fp8": 1.0, "none": 2.0}[quant]
    nbytes = sum(w * per_w for _, _, w in launches_list)
    elapsed_ms, how = 1.0, "not timed (counter merge only)"
    if time_it:
        stream = torch.cuda.current_stream()
        for fn, k, _ in launches_list:  # warm
            fn(xs[k])
        torch.cuda.synchronize()
        replay, how = None, "eager launches"
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for fn, k, _ in launches_list:
                    fn(xs[k])
            for _ in range(3):  # warm replays (clocks, caches) in front of the timed ones
                g.replay()
            torch.cuda.synchronize()
            replay, how = g.replay, "hipGraph replay of the launch list"
        except Exception:  # capture refused: time the eager launches (gaps included)
            torch.cuda.synchronize()
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            if replay is not None:
                replay()
            else:
                for fn, k, _ in launches_list:
                    fn(xs[k])
        e1.record(stream)
        torch.cuda.synchronize()
        elapsed_ms = e0.elapsed_time(e1)
    nl = len(launches_list)
    launches = iters * nl  # projections timed (a format's quantiser / finish launches ride inside their projection's share)
    avg_s = elapsed_ms * 1e-3 / (iters * nl)
    bytes_per_launch = nbytes / nl
    achieved = bytes_per_launch / avg_s
    traffic, traffic_source, traffic_parts = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if live_pmc and live_pmc.get("traffic"):
        tj = live_pmc["traffic"]
        traffic = tj.get("wgemm_bytes_per_launch")
        traffic_parts = {"fetch": tj.get("wgemm_fetch_bytes_per_launch"), "write": tj.get("wgemm_write_bytes_per_launch"),
                         "per_kernel_KB": tj.get("per_kernel_KB")}
        traffic_source = ("LIVE counters of this run: " + live_pmc["how"] + "; FETCH_SIZE x 1024 x 2 (the gfx950 correction of "
                          "MI355X_MICROARCH.md), WRITE_SIZE x 1024 uncalibrated; mean over the launch kinds of a decoder layer")
    elif quant == "int4" and solo and os.path.exists(tpath):  # the stored counters are those of the int4 engine at TP = 1 only
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("wgemm_bytes_per_launch")
            traffic_parts = {"fetch": tj.get("wgemm_fetch_bytes_per_launch"), "write": tj.get("wgemm_write_bytes_per_launch")}
            traffic_source = ("stored profile value, not a live counter: " + tj.get("summary", "profiles/pmc_traffic.json")
                              + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH x2 per the gfx950 correction, "
                                "WRITE_SIZE uncalibrated)")
        except Exception:
            traffic = None
    mfma_util = None
    spath = os.path.join(ROOT, "profiles", "pmc_sq.json")
    if live_pmc and live_pmc.get("sq"):
        mfma_util = {"source": "LIVE counters of this run: " + live_pmc["how"] + "; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                               "SQ_BUSY_CYCLES / 32)",
                     "kernels": {r["kernel"]: {"mfma_util": r.get("mfma_util"), "lds_busy": r.get("lds_busy"),
                                               "lds_bank_conflict_frac": r.get("lds_conflict_frac"),
                                               "wave_wait_frac": r.get("wave_wait_frac")} for r in live_pmc["sq"]}}
    elif os.path.exists(spath) and quant in ("int4", "smoothquant"):
        try:  # north_star's "MFMA-utilisation counters": stored values of the last tools/pmc_round.sh pass, not live counters
            sj = json.load(open(spath))
            pick = {"int4": ["wgemm4_kernel<5, 2>", "wgemm3_kernel<2, 1>", "wss_kernel<2, 4, 4, 8>", "wss_kernel<2, 2, 4, 14>"],
                    "smoothquant": ["dense8_kernel<3, 1>", "wgemm16_rows_kernel<1, 2>"]}[quant]
            mfma_util = {"source": "stored profile values (profiles/pmc_sq.json <- tools/pmc_round.sh: SQ_VALU_MFMA_BUSY_CYCLES / "
                                   "(1024 SIMDs x SQ_BUSY_CYCLES / 32), separate --pmc passes over the step's eager launches)",
                         "kernels": {k: {"mfma_util": v.get("mfma_util"), "lds_busy": v.get("lds_busy"),
                                         "lds_bank_conflict_frac": v.get("lds_conflict_frac"), "wave_wait_frac": v.get("wave_wait_frac")}
                                     for k, v in sj.items() if any(t in k for t in pick)}}
        except Exception:
            mfma_util = None
    kernel = {
        "int4": "wgemm4_kernel (row-group engine, the fused gate|up + swiglu launch; gemm_w4_v4.hip) + wgemm3_kernel (unit-loop engine, "
                "the down split-K partial launch; gemm_w4_v3.hip) + wss_kernel (short-stream engine, the q|k|v / o split-K partial "
                "launches; gemm_short.hip) -- w4a16 dequant-GEMM over pre-packed weights",
        "int8": "dense8_kernel + dense8_finish (w8a16 int8, split-K weight streaming; gemm_w8_skinny.hip)",
        "fp8": "dense8_kernel + dense8_finish (w8a16 fp8-e4m3, split-K weight streaming; gemm_w8_skinny.hip)",
        "smoothquant": "dense8_kernel (int8 x int8 MFMA, split-K planes; gemm_w8_skinny.hip) / wgemm16_rows_kernel<1, 2> (the fused gate|up "
                       "row-group loop incl. scale epilogue + swiglu; gemm_w16_rows.hip) with the per-token quantiser / scale "
                       "epilogue fused into the neighbouring launches where the step fuses them (w8a8_fused.hip): as timed here, "
                       "q|k|v = quantiser + GEMM + finish, gate|up = quantiser + GEMM + finish-swiglu, o / down = quantiser + GEMM",
        "none": "dense8_kernel 16-bit form in split-K partial mode for q|k|v, o, down (gemm_w8_skinny.hip; planes summed by the "
                "consuming norm / attention launch) + wgemm16_rows_kernel (16-bit row-group loop, gemm_w16_rows.hip) for the fused "
                "gate|up incl. its swiglu -- in-tree kernels only; averaged over the launches",
    }[quant] if batch <= 64 or quant == "none" else "wgemm_kernel (generic engine, M > 64; gemm_wq.hip)"
    return {
        "bound": "hbm", "kernel": kernel,
        "covers": "the dense linear projections (attention q|k|v / o" + ("" if getattr(model.geo, "num_experts", 0) else ", gate|up, down")
                  + "); one launch = one projection incl. its activation quantiser / finish kernels where the format has them",
        "achieved": round(achieved / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
        "frac": round(achieved / PEAK_HBM, 4), "traffic": traffic, "traffic_parts": traffic_parts, "traffic_source": traffic_source,
        "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(avg_s * 1e6, 2),
        "launches_timed": launches, "timed_as": how, "mfma_util": mfma_util,
    }


def reference_order_step(model, engine, batch, ctx, steps, warmup):
    """The drop-in route measured (round-2 review "what's weak" 8): ONE decode step written the way the reference's
    model code calls the kernel layer (models/base.py:299-319, 204-245, 81-129, 263-264) -- per layer skip_rmsnorm,
    w4a16_matmul(q), w4a16_matmul(kv), rope_emb_forward, cat, update_kv_buffer, flash_decoding, w4a16_matmul(o),
    skip_rmsnorm, w4a16_matmul(gate), w4a16_matmul(up), swiglu_forward, w4a16_matmul(down); final norm, fp16 lm_head,
    argmax -- using ONLY the 16 exported names of ``lite_llama_amd.kernels`` on the reference-format parameters
    (no merged launches, no pre-packed stream requested by the caller, no partial-sum hand-offs), captured in a hipGraph
    like the reference's CUDAGraphRunner and replayed.  The context stays at ``ctx`` tokens (the midpoint of the headline
    run: same bytes per step on average); tokens are fed back, the metadata is not advanced."""
    import torch.nn.functional as F
    import lite_llama_amd.kernels as K

    geo = model.geo
    dev = next(model.parameters()).device
    info = engine.info
    B, D = batch, geo.head_dim
    layers = []
    for layer in model.layers:
        at, mlp = layer.self_attn, layer.mlp
        # (a compacted model keeps only the decode engine's load-time layout: the REFERENCE-format tensor the drop-in caller
        # would hold is rebuilt here, bit-exact, for the duration of this measurement)
        lin = lambda m: (m.quant_method.reference_weight(m).data.contiguous(), m.weight_scale.data.contiguous(),  # noqa: E731
                         m.weight_zeros.data.contiguous(), None if m.bias is None else m.bias.data.contiguous())
        layers.append(dict(ln1=layer.input_layernorm_weight.data, ln2=layer.post_attention_layernorm_weight.data,
                           q=lin(at.q_proj), kv=lin(at.kv_proj), o=lin(at.o_proj), gate=lin(mlp.gate_proj), up=lin(mlp.up_proj),
                           down=lin(mlp.down_proj)))
    hq, hkv = model.layers[0].self_attn.num_heads, model.layers[0].self_attn.num_kv_heads
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    pos = torch.full((B, 1), ctx - 1, dtype=torch.long, device=dev)
    sel = info.b_req_tokens_table[:B, ctx - 1].contiguous()
    ids = torch.randint(0, geo.vocab_size, (B, 1), device=dev)
    scale = 1.0 / D ** 0.5
    g = 128

    def mm(x, p):
        return K.w4a16_matmul(x, p[0], p[1], p[2], group_size=g, bias=p[3])

    def step():
        h = model.embed_tokens(ids)
        cos, sin = model.rotary_emb._tables(pos, h.dtype)
        res = None
        for i, L_ in enumerate(layers):
            h, res = K.skip_rmsnorm(h, res, L_["ln1"], model.eps)
            x2 = h.view(-1, geo.hidden_size)
            xq = mm(x2, L_["q"]).view(B, hq, D)
            xkv = mm(x2, L_["kv"]).view(B, 2 * hkv, D)
            xk, xv = xkv[:, :hkv], xkv[:, hkv:]
            xq, xk = K.rope_emb_forward(xq, xk, cos, sin, B, 1)
            K.update_kv_buffer(torch.cat([xk, xv], dim=-2), sel, info.kv_buffer[i])
            kv = info.kv_buffer[i]
            att = K.flash_decoding(xq, kv[:, :hkv, :], kv[:, hkv:, :], scale, info.b_req_tokens_table, info.b_req_idx[:B], seq, ctx)
            o = mm(att.view(B, hq * D), L_["o"]).view(B, 1, geo.hidden_size)
            h, res = K.skip_rmsnorm(o, res, L_["ln2"], model.eps)
            x2 = h.view(-1, geo.hidden_size)
            a = K.swiglu_forward(mm(x2, L_["gate"]), mm(x2, L_["up"]))
            h = mm(a, L_["down"]).view(B, 1, geo.hidden_size)
        h, _ = K.skip_rmsnorm(h, res, model.norm_weight.data, model.eps)
        logits = F.linear(h, model.lm_head_weight)
        ids.copy_(torch.argmax(logits[:, -1], dim=-1, keepdim=True))

    step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del graph, layers
    torch.cuda.empty_cache()
    return {"value": round(batch / dt, 1), "unit": "tokens/s", "ms_per_step": round(dt * 1e3, 4), "ctx": ctx,
            "launches_per_layer": 13,
            "what": "the reference's decoder layer through the 16 exported kernel names only, reference-format parameters, "
                    "hipGraph replay, fixed context (drop-in route 1 of INTEGRATION.md)"}


def second_point(model, args, dev, act_dtype, geo, tp, ctx=2048):
    """The second measured point of SURVEY 8(d): decode after a 2048-token prompt (the reference's graph bucket <= 4096,
    executor/cuda_graph.py:27-28) -- 17 partitions per row: the attention's global-merge path, the fused q|k|v finished by
    the projection itself.  Same model, a fresh engine, hipGraph replays."""
    from lite_llama_amd.executor import DecodeEngine

    steps, warmup = min(args.steps, 32), 4
    eng = DecodeEngine(model, max_batch=args.batch, max_seq_len=ctx + steps + warmup + 8, device=dev, kv_dtype=act_dtype)
    first = eng.synthetic_context(args.batch, ctx, seed=11)
    marks = {}

    def on_step(i):
        if i == warmup:
            torch.cuda.synchronize()
            marks["t0"] = time.perf_counter()

    eng.decode(first, warmup + steps, use_graph=not args.no_graph, on_step=on_step)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - marks["t0"]) / steps
    w_lin, w_head, kv_bytes = algorithmic_bytes(geo, args.quant, args.batch, ctx + warmup + steps / 2, tp)
    out = {"workload": f"{args.model} {args.quant} decode, batch {args.batch}, ctx {ctx}->{ctx + warmup + steps}",
           "value": round(args.batch / dt, 1), "unit": "tokens/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "warmup": warmup,
           "step_roofline_frac_of_8TBps": round((w_lin + w_head + kv_bytes) / dt / PEAK_HBM, 4)}
    del eng
    torch.cuda.empty_cache()
    return out


def prefill_point(model, args, dev, act_dtype, geo, tp, prompt_len=512, reps=2):
    """TTFT of the reference's protocol (benchmarks/common.py:100-137 times the prefill step and the decode steps in one run):
    ``DecodeEngine.prefill`` of ``batch`` x ``prompt_len`` random token ids through the HIP path -- the > 64-row route of every
    projection + ``flash_attention2_no_pad`` (a6) + KV scatter + rope -- then the logits of the last prompt token and the greedy
    first token.  Achieved TFLOP/s = (2 x linear weights x tokens + causal attention flops) / time against the dense 16-bit
    MFMA peak (the arithmetic the dequantised weights run in)."""
    from lite_llama_amd.executor import DecodeEngine

    b = args.batch
    eng = DecodeEngine(model, max_batch=b, max_seq_len=prompt_len + 8, device=dev, kv_dtype=act_dtype)
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    ids = torch.randint(0, geo.vocab_size, (b, prompt_len), generator=g, device=dev)
    first = eng.prefill(ids)  # warm-up (first-use costs: workspaces, rotary tables, transient layouts)
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        first = eng.prefill(ids)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    tokens = b * prompt_len
    attn_w = geo.hidden_size * geo.q_size * 2 + geo.hidden_size * 2 * geo.kv_size
    if geo.num_experts:
        mlp_w = geo.num_experts_per_tok * 3 * geo.hidden_size * geo.moe_intermediate_size + geo.num_experts * geo.hidden_size
    else:
        mlp_w = 3 * geo.hidden_size * geo.intermediate_size
    lin_flops = 2.0 * geo.num_layers * (attn_w + mlp_w) / tp * tokens + 2.0 * geo.vocab_size * geo.hidden_size * b
    att_flops = geo.num_layers * b * 2.0 * 2.0 * (geo.num_heads / tp) * geo.head_dim * prompt_len * prompt_len / 2.0  # causal half
    peak = 5.0e15 if args.quant == "smoothquant" else 2.5e15  # int8 x int8 MFMA (dense) / 16-bit MFMA over widened weights
    out = {"workload": f"{args.model} {args.quant} prefill, batch {b} x {prompt_len} prompt tokens (padded grid, all rows valid)",
           "ttft_ms": round(dt * 1e3, 2), "tokens": tokens, "prefill_tokens_per_s": round(tokens / dt, 1),
           "achieved_TFLOPs": round((lin_flops + att_flops) / dt / 1e12, 1), "peak_TFLOPs": peak / 1e12,
           "frac_of_mfma_peak": round((lin_flops + att_flops) / dt / peak, 4),
           "flops": {"linear": lin_flops, "attention": att_flops}, "reps_ms": [round(t * 1e3, 2) for t in times],
           "first_tokens_checksum": int(first.long().sum().item()),
           "route": {"int4": "wgemm_prefill_kernel (M-tiled W4A16 MFMA GEMM over the load-time layout, gemm_w4_prefill.hip) + fa_prefill2",
                     "smoothquant": "w8_mtiled_kernel<3> (M-tiled int8 x int8 MFMA GEMM, gemm_w8_prefill.hip) + per-token quantiser + fa_prefill2",
                     "fp8": "w8_mtiled_kernel<1> (M-tiled fp8 -> fp16 MFMA GEMM, gemm_w8_prefill.hip) + fused_moe + fa_prefill2",
                     "int8": "w8_mtiled_kernel<2> (M-tiled int8 -> fp16 MFMA GEMM, gemm_w8_prefill.hip) + fa_prefill2",
                     "none": "library GEMM (F.linear) + fa_prefill2"}.get(args.quant, "wgemm_kernel (generic engine, gemm_wq.hip) + fa_prefill2"),
           "mfma_peak_note": ("5.0 PF = nominal dense int8 MFMA peak" if args.quant == "smoothquant" else
                              "2.5 PF = nominal dense fp16 peak at 2.4 GHz; under this load the part clocks ~1.6 GHz (DESIGN_NOTEBOOK.md 5.3)")}
    del eng
    torch.cuda.empty_cache()
    return out


SECONDARY_CONFIGS = [  # BASELINE.json configs[1], [3], [4] (configs[2] is the headline, configs[0] the CPU baseline)
    ("config 2: Qwen2.5-1.5B bf16 flash-decoding bs=32, hipGraph", ["--model", "qwen2.5-1.5b", "--quant", "none", "--dtype", "bf16", "--batch", "32"]),
    ("config 4: Llama-3-8B SmoothQuant W8A8 decode bs=32", ["--model", "llama-3-8b", "--quant", "smoothquant", "--batch", "32"]),
    ("config 5: Qwen3-30B-A3B FP8 MoE decode bs=64 (one GPU of the TP set)", ["--model", "qwen3-30b-a3b", "--quant", "fp8", "--batch", "64"]),
]


def secondary_configs(steps: int, timeout_s: float = 240.0):
    """The other GPU configurations of BASELINE.json as bounded entries of the headline line (round-4 review, item 6): each one
    is THIS script in a fresh process (its own model, engine and hipGraph; <= 32 timed steps) -- value, ms_per_step, its
    dominant-kernel ``roofline`` object and its own ``parity_check`` of one decoder layer against the CPU oracle."""
    import subprocess

    out = []
    for label, flags in SECONDARY_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), *flags, "--steps", str(min(steps, 32)), "--warmup", "4", "--as-secondary"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out.append({"config": label, "error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"})
                continue
            d = json.loads(line[-1])
            keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "graph", "roofline",
                                          "roofline_dense_projections", "parity_check", "parity_detail", "prefill")}
            keep.update({"config": label, "workload": d["config"]["workload"],
                         "step_roofline_frac_of_8TBps": d["step_roofline"]["frac_of_8TBps"],
                         "wall_seconds": round(time.perf_counter() - t0, 1)})
            out.append(keep)
        except subprocess.TimeoutExpired:
            out.append({"config": label, "error": f"no line within {timeout_s:.0f} s"})
        except Exception as exc:
            out.append({"config": label, "error": f"{type(exc).__name__}: {exc}"})
    return out


def shard_sim_points(args, degrees, timeout_s: float = 200.0):
    """The compute half of the TP scaling curve, measured on ONE GPU (round-5 review, item 5): for every degree THIS script in a
    fresh process builds rank 0's shard of the model (the shard plan's shapes, synthetic weights), replaces every all-reduce by a
    same-size local device copy (parallel_state.simulate_shard) and times the captured decode step.  Labelled for what it is."""
    import subprocess

    out = []
    for tp in degrees:
        cmd = [sys.executable, os.path.abspath(__file__), "--model", args.model, "--quant", args.quant, "--batch", str(args.batch),
               "--ctx", str(args.ctx), "--dtype", args.dtype, "--steps", str(min(args.steps, 32)), "--warmup", "4",
               "--as-shard-sim", str(tp), "--as-secondary", "--no-cpu-baseline", "--shard-sim", "none"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out.append({"tp": tp, "error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"})
                continue
            d = json.loads(line[-1])
            out.append({"tp": tp, "ms_per_step": d["ms_per_step"], "tokens_per_s_if_collectives_were_free": round(d["value"], 1),
                        "step_roofline_frac": d["step_roofline"]["frac_of_8TBps"],
                        "algorithmic_bytes_per_step": d["step_roofline"]["algorithmic_bytes_per_step_per_gpu"],
                        "shard_plan": d["config"]["shard_plan"], "graph": d["graph"]})
        except subprocess.TimeoutExpired:
            out.append({"tp": tp, "error": f"no line within {timeout_s:.0f} s"})
        except Exception as exc:
            out.append({"tp": tp, "error": f"{type(exc).__name__}: {exc}"})
    return {"what": "ONE rank's compute at the TP shard shapes on one GPU, every all-reduce replaced by a same-size local copy, "
                    "captured step, <= 32 steps: a ceiling of the scaling curve (no collective, no peers), NOT a scaling point",
            "points": out}


def pmc_probe(args):
    """--as-pmc-probe: the dense projections of THREE decoder layers of the model at its real widths (3 x 131 MB of int4 weights
    for Qwen2.5-7B: more than the 256-MB Infinity Cache, so launches meet cold weights as in the step), issued eagerly like the
    decode step issues them, four rounds.  Run under rocprofv3 --kernel-trace --pmc <counters> by live_pmc_counters()."""
    import dataclasses

    from lite_llama_amd.distributed import parallel_state as ps
    from lite_llama_amd.model import GEOMETRY, CausalLM
    from lite_llama_amd.quantization import QuantConfig

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ps.init_parallel(0, tp_size=1, dp_size=1, master_port=int(os.environ.get("MASTER_PORT", 29500)))
    geo = dataclasses.replace(GEOMETRY[args.model], num_layers=3, vocab_size=4096)
    quant = None if args.quant == "none" else QuantConfig.for_runtime_scheme(args.quant)
    with torch.device(dev):
        model = CausalLM(geo, quant)
    model.init_synthetic(seed=0, quant=quant, device=dev)
    if args.dtype == "bf16":
        model = model.to(torch.bfloat16)
    if quant is not None:
        model.compact_weights()
    launches_list, _ = projection_launches(model, args.quant)
    adt = next(model.parameters()).dtype if args.quant == "none" else torch.float16
    xs = {k: (torch.randn(args.batch, k, device=dev) * 0.5).to(adt) for _, k, _ in launches_list}
    for _ in range(4):
        for fn, k, _ in launches_list:
            fn(xs[k])
    torch.cuda.synchronize()
    print(json.dumps({"pmc_probe": "ok", "launches": 4 * len(launches_list)}), flush=True)


def live_pmc_counters(args, budget_s: float = 150.0):
    """HBM-side traffic and MFMA utilisation of the dense projections' kernels from LIVE counters: four rocprofv3 --kernel-trace
    --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets; each in its own run, as MI355X_MICROARCH.md prescribes) over
    ``bench.py --as-pmc-probe`` in a subprocess, bounded in time; ``None`` when rocprofv3 is missing, a pass fails or the budget
    runs out (the line then carries the stored profile values, labelled so)."""
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_sq
        import pmc_traffic
    except Exception:
        return None
    sets = {
        "FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
        "A": "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE".split(),
        "B": "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE".split(),
    }
    t0 = time.perf_counter()
    dbs, took = {}, {}
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("LL_LIB_OVERRIDE", None)
    with tempfile.TemporaryDirectory(prefix="ll_pmc_", dir="/tmp") as tmp:
        for tag, ctrs in sets.items():
            left = budget_s - (time.perf_counter() - t0)
            if left < 20:
                break
            out = os.path.join(tmp, tag)
            cmd = [prof, "--kernel-trace", "--pmc", *ctrs, "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--as-pmc-probe", "--model", args.model, "--quant", args.quant, "--dtype", args.dtype, "--batch", str(args.batch)]
            t1 = time.perf_counter()
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=min(left, 90.0))
            except Exception:
                continue
            took[tag] = round(time.perf_counter() - t1, 1)
            db = None
            for dirpath, _, files in os.walk(out):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(dirpath, f)
            if r.returncode == 0 and db and '"pmc_probe": "ok"' in r.stdout:
                dbs[tag] = db
        res = {"how": "rocprofv3 --kernel-trace --pmc <one counter set per pass> over `bench.py --as-pmc-probe` (the projections of three "
                      "decoder layers at the model's widths, eager launches, batch %d) in subprocesses of this run; passes (s): %s"
                      % (args.batch, json.dumps(took))}
        gemms = ["wgemm", "wss_kernel", "dense8", "dense_ss"]
        try:
            if "FETCH_SIZE" in dbs and "WRITE_SIZE" in dbs and args.quant == "int4":
                gf = {k: v for k, v in pmc_traffic.per_kernel(dbs["FETCH_SIZE"], "FETCH_SIZE", gemms).items()}
                gw = {k: v for k, v in pmc_traffic.per_kernel(dbs["WRITE_SIZE"], "WRITE_SIZE", gemms).items()}
                if gf and gw:
                    res["traffic"] = pmc_traffic.summarise(gf, gw, {}, {}, "live_run")
            if "A" in dbs and "B" in dbs:
                rows = pmc_sq.rows_from(pmc_sq.load(dbs["A"]), pmc_sq.load(dbs["B"]), gemms + ["wgemm16_rows"])
                res["sq"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
        except Exception as exc:
            res["error"] = f"{type(exc).__name__}: {exc}"
    return res if ("traffic" in res or "sq" in res) else None


def moe_roofline(model, batch, iters=6):
    """Config 5's dominant kernel is the grouped expert GEMM (moe_gemm_kernel2), not the attention projections: time the
    routed block of every layer (moe_align + gate|up GEMM + silu_and_mul + down GEMM + moe_sum) on random routing and price
    it against the expert bytes the batch touches (n_e(B) experts x 3 x H x I bytes per layer)."""
    from lite_llama_amd.model import SparseMoeBlock

    blocks = [m for m in model.modules() if isinstance(m, SparseMoeBlock)]
    if not blocks:
        return None
    geo = model.geo
    dev = next(model.parameters()).device
    x = torch.randn(batch, geo.hidden_size, device=dev, dtype=torch.float16) * 0.5
    routes = [b._route(x) for b in blocks[:1]][0]
    for b in blocks[:2]:
        b.quant_method.apply(b, x, routes[0].half(), routes[1])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for b in blocks:
            b.quant_method.apply(b, x, routes[0].half(), routes[1])
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / (iters * len(blocks))
    hit = int(torch.unique(routes[1]).numel())
    nbytes = hit * 3 * geo.hidden_size * blocks[0].moe_intermediate_size * 1.0
    return {"bound": "hbm", "kernel": "moe_gemm_kernel2 x2 inside the routed block (moe_align + grouped gate|up GEMM + silu_and_mul + "
                                      "grouped down GEMM + moe_sum; moe.hip), eager launches",
            "achieved": round(nbytes / avg_s / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
            "frac": round(nbytes / avg_s / PEAK_HBM, 4), "traffic": None, "bytes_per_block": int(nbytes),
            "experts_hit": hit, "avg_block_us": round(avg_s * 1e6, 2), "blocks_timed": iters * len(blocks)}


def measured_copy_bandwidth(dev, nbytes=1 << 30, iters=10):
    """Device-to-device copy of a 1-GiB buffer (read + write bytes per second) -- the achievable HBM rate on THIS box,
    printed next to the nominal peak the roofline divides by."""
    a = torch.empty(nbytes // 2, dtype=torch.float16, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3)


def parity_check(geo, quant_name, sample):
    """The exact launch sequence the timed step makes for ONE decoder layer of the headline workload (fused q|k|v as
    split-K partials -> one-launch attention -> o partials -> add-and-normalise over partials -> gate|up + swiglu -> down
    partials -> norm -> lm_head), on the weights / K,V / tokens of the CPU oracle sample, against the oracle's logits.
    Outside the timed region; the oracle is the checker here, never the thing measured."""
    import types
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    p, layers = sample["params"], sample["layers"]
    g1 = tiny_geometry(name=geo.name + "-parity", hidden_size=geo.hidden_size, intermediate_size=geo.intermediate_size,
                       num_layers=layers, num_heads=geo.num_heads, num_kv_heads=geo.num_kv_heads, head_dim=geo.head_dim,
                       vocab_size=geo.vocab_size, rope_theta=geo.rope_theta, rms_norm_eps=geo.rms_norm_eps, qkv_bias=geo.qkv_bias,
                       use_qk_norm=geo.use_qk_norm, num_experts=geo.num_experts, num_experts_per_tok=geo.num_experts_per_tok,
                       moe_intermediate_size=geo.moe_intermediate_size, norm_topk_prob=geo.norm_topk_prob)
    m = CausalLM(g1)
    m.load_state_dict(p, strict=True)
    m = m.to("cuda")
    if quant_name != "none":
        m.quantize_(QuantConfig.for_runtime_scheme(quant_name))
    info_c = sample["info"]
    ctx = int(sample["pos"][0, 0])
    m.rotary_emb.ensure(ctx + 8, "cuda")
    kv = [k.clone().cuda() for k in sample["kv_before"]]
    info = types.SimpleNamespace(kv_buffer=kv, cur_select_index=info_c.cur_select_index.cuda(),
                                 b_req_tokens_table=info_c.b_req_tokens_table.cuda(), b_start_loc=None,
                                 b_req_idx=info_c.b_req_idx.cuda(), b_seq_len=info_c.b_seq_len.cuda(),
                                 max_actual_seq_len=info_c.max_actual_seq_len)
    with torch.no_grad():
        got = m(sample["ids"].cuda(), sample["pos"].cuda(), info).float().cpu()
    ref = sample["logits"].float()
    err = (got - ref).abs()
    # smoothquant: the reference's own tolerance for ONE W8A8 projection is 1e-1 (tests/kernels/test_quantization.py); the
    # layer chains four of them behind a dynamic per-token quantiser whose truncation turns 1-ulp input differences into
    # whole-code differences -- measured 2.5 - 3 % relative rms per row at Llama-3-8B widths, i.e. ~6 sigma at 2e-1 over 4 M logits
    tol = 2e-1 if quant_name == "smoothquant" else 3e-2
    row_ok = torch.all((err <= tol + tol * ref.abs()).flatten(1), dim=1)
    tie_note = None
    margin = sample.get("route_margin")
    if margin is not None and not bool(row_ok.all()):
        # MoE: a row whose 8th and 9th router probabilities differ by less than fp16 rounding of the router logits may pick
        # another expert on the device than in the oracle -- a discontinuity of top-k, not an arithmetic difference.  Such rows
        # are reported and left out of the verdict; every other row must pass.
        tie = margin < 4e-3
        tie_note = {"rows_failing": int((~row_ok).sum()), "of_which_router_near_ties": int((~row_ok & tie).sum()),
                    "near_tie_rows_total": int(tie.sum()), "near_tie_rule": "relative gap between the k-th and (k+1)-th router probability < 4e-3 in the oracle"}
        excused = int((~row_ok & tie).sum())
        tie_note["excused_cap"] = max(1, got.shape[0] // 16)   #  more excused rows than this is a FAILURE, not a tie
        if excused <= tie_note["excused_cap"]:
            row_ok = row_ok | tie
    ok = bool(row_ok.all())
    strict = None
    if quant_name == "smoothquant":  # second verdict at the reference's single-projection tolerance (reported, not the gate)
        strict = {"tolerance": "1e-1 + 1e-1 |ref|", "rows_passing": int(torch.all((err <= 1e-1 + 1e-1 * ref.abs()).flatten(1), dim=1).sum()),
                  "rows": int(got.shape[0]), "note": "the reference's 1e-1 is its tolerance for ONE W8A8 projection; the layer chains four behind a "
                                                     "truncating per-token quantiser (whole-code flips from 1-ulp input differences)"}
    rel_rms = float(((got - ref).flatten(1).pow(2).mean(1).sqrt() / ref.flatten(1).pow(2).mean(1).sqrt().clamp_min(1e-9)).max())
    rows = info_c.cur_select_index.long()
    kv_got, kv_ref = kv[0][rows.cuda()].float().cpu(), sample["kv_after"][0][rows].float()
    kv_err = float((kv_got - kv_ref).abs().max())
    kv_ok = bool(torch.all((kv_got - kv_ref).abs() <= 2e-2 + 2e-2 * kv_ref.abs()))  # rope / KV rows at 2e-2
    same_tok = int((got[:, -1].argmax(-1) == ref[:, -1].argmax(-1)).sum())
    del m
    torch.cuda.empty_cache()
    return {"ok": ok and kv_ok, "tolerance": f"|got - ref| <= {tol} + {tol} |ref| (logits), 2e-2 + 2e-2 |ref| (new K/V rows)", "max_abs_err_logits": round(float(err.max()), 5),
            "max_row_relative_rms_err_logits": round(rel_rms, 5), "router_ties": tie_note, "at_reference_projection_tolerance": strict,
            "max_abs_err_new_kv_rows": round(kv_err, 5), "argmax_agree": f"{same_tok}/{got.shape[0]}",
            "what": f"{layers} decoder layer(s) + final norm + lm_head at the headline shape (batch {got.shape[0]}, ctx {ctx}), "
                    "HIP step vs CPU oracle on identical weights / K,V / tokens"}


def cpu_baseline_protocol(gen_len=256, prompt_len=32, threads=None):
    """The reference's CPU-runnable case (BASELINE.json configs[0]): benchmarks/bench_hf_baseline.py through
    benchmarks/common.py::HFBackend.measure (common.py:174-221) restated for the host cores -- Qwen2.5-0.5B geometry
    (random weights: no checkpoints offline) in the Hugging Face transformers implementation (sdpa attention, fp32 on
    CPU), batch 1, greedy with min_new_tokens == max_new_tokens, two 8-token warm-up generations, TTFT from a
    one-token generation, TPOT = (total - TTFT) / (steps - 1), tokens/s = generated tokens / total.  The prompt is
    ``prompt_len`` random token ids (the reference's text prompts need its tokenizer files)."""
    import platform
    from transformers import AutoModelForCausalLM, Qwen2Config

    cfg = Qwen2Config(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=24,
                      num_attention_heads=14, num_key_value_heads=2, max_position_embeddings=32768, rope_theta=1e6,
                      rms_norm_eps=1e-6, tie_word_embeddings=True, attn_implementation="sdpa")
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(cfg).float().eval()
    ids = torch.randint(0, cfg.vocab_size, (1, prompt_len))
    att = torch.ones_like(ids)
    kw = dict(do_sample=False, pad_token_id=0)
    all_threads = torch.get_num_threads()
    with torch.no_grad():
        # thread count: a 0.5B model at batch 1 does not scale to every core of a big host (128 threads run it ~15x
        # SLOWER than 16 on the 256-core EPYC of the GPU box); the protocol's "all threads" is therefore replaced by the
        # best of a short probe over {all, 32, 16, 8} threads, and the count used is reported in "cores"
        best = None
        for nt in ([threads] if threads else sorted({all_threads, 32, 16, 8}, reverse=True)):
            if nt > all_threads:
                continue
            torch.set_num_threads(nt)
            model.generate(ids, attention_mask=att, min_new_tokens=2, max_new_tokens=2, **kw)
            t0 = time.perf_counter()
            model.generate(ids, attention_mask=att, min_new_tokens=4, max_new_tokens=4, **kw)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        torch.set_num_threads(best[1])
        # bounded sample: the protocol's 256 tokens unless that would take much longer than ~30 s on this host
        gen_len = max(16, min(gen_len, int(30.0 / max(best[0] / 4, 1e-3))))
        for _ in range(2):
            model.generate(ids, attention_mask=att, min_new_tokens=8, max_new_tokens=8, **kw)
        t0 = time.perf_counter()
        model.generate(ids, attention_mask=att, min_new_tokens=1, max_new_tokens=1, **kw)
        ttft = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = model.generate(ids, attention_mask=att, min_new_tokens=gen_len, max_new_tokens=gen_len, **kw)
        total = time.perf_counter() - t0
    steps = out.shape[1] - prompt_len
    cpu = platform.processor() or platform.machine()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    used = torch.get_num_threads()
    torch.set_num_threads(all_threads)
    return {"value": round(steps / total, 2), "unit": "tokens/s", "cores": used, "kind": "port",
            "sample": f"bench_hf_baseline.py protocol (common.py:174-221) on the host: Qwen2.5-0.5B geometry, random weights, "
                      f"HF transformers sdpa fp32, batch 1, greedy, prompt {prompt_len} ids, 2 x 8-token warm-up, {steps} generated tokens",
            "ttft_ms": round(ttft * 1e3, 1), "tpot_ms": round((total - ttft) / max(steps - 1, 1) * 1e3, 2),
            "sample_seconds": round(total + ttft, 1), "cpu_model": cpu, "host_cores": os.cpu_count()}


def cpu_baseline(geo, batch, ctx, layers, quant):
    """The CPU oracle (a port of the reference algorithm, oracle/model.py) on a bounded sample:
    ``layers`` decoder layers + final norm + lm_head of the same decode step (same batch / context
    / geometry / int4 format), extrapolated to the full depth."""
    from oracle import oracle as O
    from oracle.model import OracleModel
    import types

    torch.manual_seed(0)
    H, I, HQ, HKV, D, V = geo.hidden_size, geo.intermediate_size, geo.num_heads, geo.num_kv_heads, geo.head_dim, geo.vocab_size
    p = {"embed_tokens.weight": (torch.randn(V, H) * 0.02).half(), "lm_head_weight": (torch.randn(V, H) * 0.02).half(),
         "norm_weight": torch.ones(H).half()}
    dense = {"self_attn.q_proj": (HQ * D, H), "self_attn.kv_proj": (2 * HKV * D, H), "self_attn.o_proj": (H, HQ * D)}
    if not geo.num_experts:
        dense.update({"mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)})
    for li in range(layers):
        pre = f"layers.{li}."
        p[pre + "input_layernorm_weight"] = torch.ones(H).half()
        p[pre + "post_attention_layernorm_weight"] = torch.ones(H).half()
        for name, (n, k) in dense.items():
            p[pre + name + ".weight"] = (torch.randn(n, k) * 0.02).half()
        if geo.qkv_bias:
            p[pre + "self_attn.q_proj.bias"] = torch.zeros(HQ * D).half()
            p[pre + "self_attn.kv_proj.bias"] = torch.zeros(2 * HKV * D).half()
        if geo.use_qk_norm:
            p[pre + "self_attn.q_norm_weight"] = (1 + 0.1 * torch.randn(D)).half()
            p[pre + "self_attn.k_norm_weight"] = (1 + 0.1 * torch.randn(D)).half()
        if geo.num_experts:
            E, MI = geo.num_experts, geo.moe_intermediate_size
            p[pre + "mlp.gate_weight"] = (torch.randn(E, H) * 0.05).half()
            p[pre + "mlp.experts.gate_up_proj"] = (torch.randn(E, 2 * MI, H) * 0.02).half()
            p[pre + "mlp.experts.down_proj"] = (torch.randn(E, H, MI) * 0.02).half()
    moe = (geo.num_experts, geo.num_experts_per_tok, geo.moe_intermediate_size, geo.norm_topk_prob) if geo.num_experts else None
    om = OracleModel(p, H, I, layers, HQ, HKV, D, V, eps=geo.rms_norm_eps, rope_theta=geo.rope_theta,
                     quant=None if quant == "none" else quant, qk_norm=geo.use_qk_norm, moe=moe)
    # quantise outside the timed region
    for li in range(layers):
        for name in dense:
            key = f"layers.{li}.{name}.weight"
            om._linear(torch.zeros(1, p[key].shape[1]).half(), key)
    rows = batch * (ctx + 1)
    kv = [(torch.randn(rows, 2 * HKV, D) * 0.5).half() for _ in range(layers)]
    table = torch.arange(rows, dtype=torch.int32).view(batch, ctx + 1)
    info = types.SimpleNamespace(kv_buffer=kv, cur_select_index=table[:, ctx].contiguous(), b_req_tokens_table=table,
                                 b_start_loc=None, b_req_idx=torch.arange(batch, dtype=torch.int32),
                                 b_seq_len=torch.full((batch,), ctx + 1, dtype=torch.int32), max_actual_seq_len=ctx + 1)
    ids = torch.randint(0, V, (batch, 1))
    pos = torch.full((batch, 1), ctx)
    kv_before = [k.clone() for k in kv]
    margins = []
    if moe is not None:  # the relative gap between the k-th and (k+1)-th router probability of every row (see parity_check)
        real_block = om._moe_block

        def block(x2, pre):
            pr = torch.softmax((x2.float() @ p[pre + "mlp.gate_weight"].float().T).to(x2.dtype), dim=-1, dtype=torch.float32)
            top = torch.topk(pr, moe[1] + 1, dim=-1).values
            margins.append((top[:, -2] - top[:, -1]) / top[:, -2])
            return real_block(x2, pre)

        om._moe_block = block
    t0 = time.perf_counter()
    logits = om.forward(ids, pos, info)
    O.greedy_argmax(logits[:, -1])
    t_all = time.perf_counter() - t0
    # time the non-layer part (embedding + final norm + lm_head + argmax) to extrapolate honestly
    om0 = OracleModel(p, H, I, 0, HQ, HKV, D, V, eps=geo.rms_norm_eps, rope_theta=geo.rope_theta, qk_norm=geo.use_qk_norm, moe=moe)
    t0 = time.perf_counter()
    O.greedy_argmax(om0.forward(ids, pos, info)[:, -1])
    t_head = time.perf_counter() - t0
    per_layer = max(t_all - t_head, 1e-9) / layers
    step_s = per_layer * geo.num_layers + t_head
    sample = {"params": p, "layers": layers, "info": info, "ids": ids, "pos": pos, "logits": logits, "kv_before": kv_before,
              "kv_after": kv, "route_margin": torch.stack(margins).min(0).values if margins else None}
    return {"value": round(batch / step_s, 2), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle decode step, batch {batch}, ctx {ctx}: {layers} of {geo.num_layers} layers timed "
                      f"({per_layer:.2f} s/layer) + lm_head/argmax ({t_head:.2f} s), extrapolated to full depth",
            "sample_seconds": round(t_all + t_head, 1)}, sample


def choose_allreduce(ps, elems, dev, strict):
    """Enable the one-shot peer-to-peer all-reduce (csrc/tp_allreduce.hip) if -- on THIS hardware, now -- it reproduces
    RCCL's sums on random payloads of the step's size; every rank takes the same decision (MIN over the group).  Returns
    the label that goes into the JSON line.  ``strict``: a failed check raises instead of falling back."""
    import torch.distributed as dist

    why = ""
    ok = 1
    try:
        ps.enable_oneshot_all_reduce(elems)
        for it in range(4):
            g = torch.Generator(device=dev).manual_seed(1234 + 17 * it + ps.get_tp_rank())
            x = (torch.randn(elems, device=dev, generator=g) * 0.25).half()
            y = x.clone()
            ps._ONESHOT.all_reduce(x)
            dist.all_reduce(y, group=ps._TP_GROUP)
            torch.cuda.synchronize()
            # RCCL adds fp16 values hop by hop, the kernel adds in fp32 and rounds once: equal up to that rounding
            if not torch.allclose(x.float(), y.float(), rtol=2e-2, atol=2e-3):
                ok, why = 0, "sums differ from RCCL's"
                break
        if ok and ps.oneshot_error():
            ok, why = 0, "a peer flag timed out"
    except Exception as exc:  # mapping refused, allocation failed, ...
        ok, why = 0, f"{type(exc).__name__}: {exc}"[:200]
    # every rank learns every rank's verdict AND the first failing rank's reason (parallel_state.collective_decision)
    use, label, reason = ps.collective_decision(ok, why, group=ps._TP_GROUP)
    if use:
        return f"oneshot (peer-mapped buffers, checked against the {ps._backend()} collective at start-up)", ""
    if ps._ONESHOT is not None:
        ps._ONESHOT.close()
        ps._ONESHOT = None
    if strict:
        raise SystemExit(f"--allreduce oneshot: start-up check failed ({reason})")
    print(f"[bench] one-shot all-reduce not used ({reason}); RCCL carries the collective", file=sys.stderr)
    return label, reason


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves (the reference's TP entry point spawns
    its workers itself too, lite_llama/cli.py:37-112, 397-476) -- one process per GPU under torch.distributed.run on a free
    local port, stdout / stderr passed through, exit status = the job's.  On a box with fewer than N devices the ranks share
    device 0 (debugging set-up: collectives staged through gloo or carried by the one-shot kernel over IPC mappings); the
    JSON line says so in ``config.parallelism_note``."""
    import socket
    import subprocess

    env = dict(os.environ)
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        env.setdefault("LL_BENCH_DEVICE", "0")
        env.setdefault("LL_DIST_BACKEND", "gloo")
        env["LL_BENCH_SHARED_DEVICE"] = f"{args.gpus} ranks share {max(ndev, 1)} device(s): debugging set-up, not a scaling point"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.as_secondary:
        args.no_secondary = True
        if args.quant not in ("smoothquant", "fp8", "int8"):  # the 8-bit configurations carry their own TTFT (round-5 review, item 7)
            args.no_prefill = True
    if args.as_pmc_probe:
        pmc_probe(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    # LL_BENCH_DEVICE: debugging knob -- all ranks on one device (with LL_DIST_BACKEND=gloo: RCCL refuses that), to run
    # the multi-rank code path on a one-GPU box
    local_rank = int(os.environ.get("LL_BENCH_DEVICE", os.environ.get("LOCAL_RANK", 0)))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from lite_llama_amd.distributed import parallel_state as ps
    from lite_llama_amd.executor import DecodeEngine
    from lite_llama_amd.model import GEOMETRY, CausalLM
    from lite_llama_amd.quantization import QuantConfig

    geo = GEOMETRY[args.model]
    # TP degree: the largest divisor of the GPU count that has a shard plan -- the reference's equal cuts where its rules
    # hold (tp | Hq, tp | Hkv, shards multiples of the scale group), else the extension plan of distributed/partition.py
    # (KV heads replicated for tp > Hkv, whole scale groups per rank): Qwen2.5-7B runs TP = 8 on it.  MoE geometries keep
    # the reference rule.  Remaining GPUs are data-parallel replicas (no collective between them).
    from lite_llama_amd.distributed.partition import admissible_tp, make_plan, scale_unit
    if geo.num_experts:
        # MoE: the attention heads on a plan (KV heads replicated for tp > Hkv), the experts on the reference's equal cut of their
        # intermediate dimension -- admitted when the cut keeps whole scale blocks of the run's quantisation (per-channel
        # formats: any cut; 128 x 128 blocks: multiples of 128)
        q_run = None if args.quant == "none" else QuantConfig.for_runtime_scheme(args.quant)
        tp, plan_note = 1, None
        for cand in (8, 4, 2, 1):
            if world % cand or geo.moe_intermediate_size % cand:
                continue
            if q_run is not None and not q_run.shard_is_aligned(geo.moe_intermediate_size // cand):
                continue
            try:
                plan_note = make_plan(geo.num_heads, geo.num_kv_heads, geo.head_dim, cand * 128, cand).describe() + \
                    f"; experts: {geo.moe_intermediate_size // cand} intermediate channels per rank"
            except ValueError:
                continue
            tp = cand
            break
    else:
        unit = scale_unit(None if args.quant == "none" else QuantConfig.for_runtime_scheme(args.quant), geo.intermediate_size)
        tp = admissible_tp(geo.num_heads, geo.num_kv_heads, geo.head_dim, geo.intermediate_size, world, unit)  # the model's own rule
        plan_note = make_plan(geo.num_heads, geo.num_kv_heads, geo.head_dim, geo.intermediate_size, tp, unit).describe()
    if args.parallelism == "dp" and not args.as_shard_sim:
        tp, plan_note = 1, "replicas only (--parallelism dp): every rank runs the whole one-GPU step on its own batch, no collective"
    if args.as_shard_sim > 1:
        if world != 1:
            raise SystemExit("--as-shard-sim is a one-process mode")
        tp = args.as_shard_sim
        if geo.num_experts:
            plan_note = make_plan(geo.num_heads, geo.num_kv_heads, geo.head_dim, tp * 128, tp).describe() + \
                f"; experts: {geo.moe_intermediate_size // tp} intermediate channels per rank"
        else:
            unit = scale_unit(None if args.quant == "none" else QuantConfig.for_runtime_scheme(args.quant), geo.intermediate_size)
            plan_note = make_plan(geo.num_heads, geo.num_kv_heads, geo.head_dim, geo.intermediate_size, tp, unit).describe()
        plan_note = f"rank 0 of tp{tp}, collectives simulated by local copies: " + plan_note
        dp = 1
        ps.simulate_shard(tp, 0)
    else:
        dp = world // tp
        ps.init_parallel(rank, tp_size=tp, dp_size=dp, master_port=int(os.environ.get("MASTER_PORT", 29500)))
    allreduce_how, allreduce_reason = ("none (tp1)" if tp == 1 else "rccl"), ("" if tp == 1 else "--allreduce rccl")
    if args.as_shard_sim > 1:
        allreduce_how, allreduce_reason = "simulated: same-size local device copy (no peers)", "--as-shard-sim"
    elif tp > 1 and args.allreduce != "rccl":
        allreduce_how, allreduce_reason = choose_allreduce(ps, args.batch * geo.hidden_size, dev, strict=args.allreduce == "oneshot")
    if world > 1 and not torch.distributed.is_initialized():  # pure DP: still need the timing barrier
        kw = {"device_id": dev} if ps._backend() == "nccl" else {}
        torch.distributed.init_process_group(ps._backend(), rank=rank, world_size=world, **kw)
    devices = None
    if world > 1:  # one rank per GPU when the box has the GPUs (a launcher pinning every rank to device 0 must not pass for a scaling point)
        devices = ps.assert_distinct_devices(local_rank, world)

    quant = None if args.quant == "none" else QuantConfig.for_runtime_scheme(args.quant)
    t_build = time.perf_counter()
    with torch.device(dev):
        model = CausalLM(geo, quant)
    model.init_synthetic(seed=0, quant=quant, device=dev)
    act_dtype = torch.float16
    if args.dtype == "bf16":
        if quant is not None:
            raise SystemExit("--dtype bf16 needs --quant none (the quantised GEMMs follow the reference: fp16 activations only)")
        model = model.to(torch.bfloat16)
        act_dtype = torch.bfloat16
    released = 0
    if quant is not None and not args.keep_reference_weights:
        released = model.compact_weights()  # int4: the load-time layout becomes the only resident copy of the weights
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    total = args.warmup + args.steps
    engine = DecodeEngine(model, max_batch=args.batch, max_seq_len=args.ctx + total + 8, device=dev,
                              kv_block_size=args.kv_block_size or None, kv_dtype=act_dtype)
    first = engine.synthetic_context(args.batch, args.ctx, seed=1 + ps.get_dp_rank(), scattered=args.scattered)

    marks = {}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def on_step(i):
        if i == args.warmup:
            barrier()
            marks["t0"] = time.perf_counter()

    use_graph = not args.no_graph
    if use_graph and tp > 1 and os.environ.get("LL_DIST_BACKEND") == "gloo" and not allreduce_how.startswith("oneshot"):
        # debugging set-up (ranks sharing a device, collectives staged through the host): a gloo collective cannot be
        # recorded, and a refused capture is not recoverable on this stack -- measure eager launches and say so
        use_graph = False
        print("[bench] gloo collectives cannot be captured: measuring eager launches", file=sys.stderr, flush=True)
    graph_note = "hipGraph" if use_graph else "eager (--no-graph or uncapturable backend)"
    graph_error = None
    try:
        if args.steps_per_graph > 1 and (args.warmup % args.steps_per_graph or args.steps % args.steps_per_graph):
            raise SystemExit("--steps-per-graph must divide --warmup and --steps (the timed region starts on a replay boundary)")
        out = engine.decode(first, total, use_graph=use_graph, on_step=on_step, steps_per_graph=args.steps_per_graph)
    except Exception as exc:
        if not use_graph:
            raise
        graph_error = f"{type(exc).__name__}: {exc}"
        out = None
    if use_graph and world > 1:
        # every rank must measure the same thing: one refusal turns the whole job eager (and the line says so)
        flag = torch.tensor([0 if graph_error else 1], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if int(flag.item()) == 0 and graph_error is None:
            graph_error = "graph capture failed on another rank"
    if graph_error is not None:
        # NOT silent: the JSON line carries "graph": false and the reason; stderr too
        print(f"[bench] hipGraph capture FAILED ({graph_error}); measuring eager launches", file=sys.stderr, flush=True)
        use_graph = False
        graph_note = "eager (graph capture failed)"
        marks.clear()
        engine = DecodeEngine(model, max_batch=args.batch, max_seq_len=args.ctx + total + 8, device=dev,
                              kv_block_size=args.kv_block_size or None, kv_dtype=act_dtype)
        first = engine.synthetic_context(args.batch, args.ctx, seed=1 + ps.get_dp_rank(), scattered=args.scattered)
        out = engine.decode(first, total, use_graph=False, on_step=on_step)
    barrier()
    elapsed = time.perf_counter() - marks["t0"]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape == (args.batch, total)

    global_batch = args.batch * dp
    tokens = global_batch * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    ctx_mid = args.ctx + args.warmup + args.steps / 2
    w_lin, w_head, kv_bytes = algorithmic_bytes(geo, args.quant, args.batch, ctx_mid, tp)
    step_bytes = w_lin + w_head + kv_bytes

    result = {
        "metric": "decode tokens/s (Qwen2.5-7B W4A16 g128, batch 64, greedy, ctx 512+)" if args.model == "qwen2.5-7b"
        else f"decode tokens/s ({args.model} {args.quant}, batch {args.batch})",
        "value": round(tokens / elapsed, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if dp == 1 else "weak", "vs_baseline": None,
        "dtype": {"int4": "f16 (int4 weights, fp32 accumulate)", "int8": "f16 (int8 weights, fp32 accumulate)",
                  "fp8": "f16 (fp8-e4m3 weights, fp32 accumulate)", "smoothquant": "int8 (int32 accumulate, f16 epilogue)",
                  "none": f"{args.dtype} (fp32 accumulate)"}[args.quant],
        "data": "synthetic", "graph": bool(use_graph), "graph_error": graph_error,
        "allreduce_error": ps.oneshot_error(),  # 0, or 1: a peer flag of the one-shot all-reduce timed out during the run
        "config": {"workload": f"{args.model} {args.quant} decode, batch {args.batch}/replica, ctx {args.ctx}->"
                               f"{args.ctx + total}, {graph_note}" + (", scattered KV rows" if args.scattered else "")
                               + (f", KV paged in blocks of {args.kv_block_size}" if args.kv_block_size else ""),
                   "global_batch": global_batch,
                   "parallelism": f"dp{dp}xtp{tp}", "allreduce": allreduce_how, "allreduce_fallback_reason": allreduce_reason or None,
                   "ranks": world, "rank_devices": None if devices is None else [f"{h}:{d}" for h, d, _ in devices],
                   "collective_backend": ("none" if world == 1 else ps._backend() + (" (= RCCL)" if ps._backend() == "nccl" else "")),
                   "shard_plan": plan_note, "steps_per_graph_launch": args.steps_per_graph if use_graph else None,
                   "parallelism_note": os.environ.get("LL_BENCH_SHARED_DEVICE"), "build_seconds": round(t_build, 1)},
        "step_roofline": {"algorithmic_bytes_per_step_per_gpu": int(step_bytes),
                          "achieved_GBps_per_gpu": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                          "frac_of_8TBps": round(step_bytes / (elapsed / args.steps) / PEAK_HBM, 4)},
    }
    w_lin_alg, w_head_alg, _ = algorithmic_bytes(geo, args.quant, args.batch, 0, tp)
    result["memory"] = {
        "weights_resident_bytes": int(model.weight_bytes()),   # every distinct parameter / load-time-layout storage of this rank
        "weights_algorithmic_bytes": int(w_lin_alg + w_head_alg),  # linear weights in reference format + fp16 lm_head
        "reference_layout_released_bytes": int(released),
        "kv_pool_bytes": int(sum(t.numel() * t.element_size() for t in engine.info.kv_buffer)),
        "allocated_after_warmup_bytes": int(torch.cuda.memory_allocated()),
        "note": "resident = embedding + norms + biases + the int4 load-time layout (0.5 B / weight) + the fp32 scale / zero grids "
                "and their fp16 pair layout (2 x 0.0625 B / weight) + fp16 lm_head",
    }
    if rank == 0:
        if world == 1 and not args.no_secondary and not args.scattered and not args.kv_block_size and args.ctx != 2048:
            try:  # "report a second point at prompt_len = 2048" -- same model, its own engine, fewer steps
                result["secondary"] = [second_point(model, args, dev, act_dtype, geo, tp)]
            except Exception as exc:
                result["secondary"] = [{"error": f"{type(exc).__name__}: {exc}"}]
        if world == 1 and not args.no_prefill:
            try:  # TTFT of the same workload through the HIP path (the reference times it in the same run)
                result["prefill"] = prefill_point(model, args, dev, act_dtype, geo, tp, prompt_len=min(args.ctx, 512))
            except Exception as exc:
                result["prefill"] = {"error": f"{type(exc).__name__}: {exc}"}
        live_wanted = (world == 1 and not args.no_live_pmc and not args.as_secondary and not args.as_shard_sim and not geo.num_experts
                       and args.quant in ("int4", "smoothquant") and args.batch <= 64)
        # the launches are timed FIRST, on the warm chip (the counter passes below leave this process idle for ~12 s: a timing taken
        # after them starts on a chip that has dropped its clocks); the counters of the run are merged into the object afterwards
        rf = gemm_roofline(model, args.batch, args.quant)
        if live_wanted and rf is not None:
            try:  # round-5 review, "weak" 10: counters of THIS run next to the timed launches (stored values only as a fallback)
                live = live_pmc_counters(args)
                if live:
                    rf_live = gemm_roofline(model, args.batch, args.quant, live_pmc=live, time_it=False)
                    rf.update({k: rf_live[k] for k in ("traffic", "traffic_parts", "traffic_source", "mfma_util")})
            except Exception:
                pass
        if geo.num_experts and quant is not None:
            try:  # the dominant kernel of a MoE model is the grouped expert GEMM; the dense projections stay as a second object
                mrf = moe_roofline(model, args.batch)
            except Exception as exc:
                mrf = {"error": f"{type(exc).__name__}: {exc}"}
            result["roofline_dense_projections"] = rf
            rf = mrf
        result["roofline"] = rf
        headline_line = (world == 1 and args.model == "qwen2.5-7b" and args.quant == "int4" and not args.as_secondary
                         and not args.scattered and not args.kv_block_size and not args.as_shard_sim)
        if (args.reference_order or (headline_line and not args.no_reference_order)) and world == 1 and args.quant == "int4" \
                and not geo.num_experts and not args.as_shard_sim:
            try:  # the drop-in route next to the headline (round-5 review, item 6): bounded unless asked for explicitly
                ro_steps, ro_warm = (args.steps, args.warmup) if args.reference_order else (min(args.steps, 20), 4)
                result["reference_order"] = reference_order_step(model, engine, args.batch, int(ctx_mid), ro_steps, ro_warm)
                result["reference_order"]["vs_headline_ms_per_step"] = round(result["reference_order"]["ms_per_step"] / ms_per_step, 3)
            except Exception as exc:
                result["reference_order"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            d2d = measured_copy_bandwidth(dev)
            result["hbm"] = {"peak_GBps_nominal": PEAK_HBM / 1e9, "measured_d2d_copy_GBps": round(d2d / 1e9, 1),
                             "note": "1-GiB device-to-device copy, read + write bytes; roofline fractions divide by the nominal peak"}
        except Exception as exc:
            result["hbm"] = {"peak_GBps_nominal": PEAK_HBM / 1e9, "error": f"{type(exc).__name__}: {exc}"}
        sim = args.shard_sim if args.shard_sim is not None else ("2,4,8" if headline_line and not args.no_secondary else "none")
        run_secondary_configs = (world == 1 and not args.no_secondary and not args.no_secondary_configs and args.model == "qwen2.5-7b"
                                 and args.quant == "int4" and not args.scattered and not args.kv_block_size and not args.as_shard_sim)
        if run_secondary_configs or (sim != "none" and world == 1 and not args.as_shard_sim):
            del engine
            torch.cuda.empty_cache()
        if sim != "none" and world == 1 and not args.as_shard_sim:
            result["shard_sim"] = shard_sim_points(args, [int(t) for t in sim.split(",") if t.strip()])
        if run_secondary_configs:
            result.setdefault("secondary", []).extend(secondary_configs(args.steps))
        if world == 1 and not args.no_cpu_baseline:
            if args.as_secondary:
                result["cpu_baseline"] = {"note": "secondary entry: the host baselines ride on the headline line"}
            else:
                try:  # SURVEY 8(d) / BASELINE configs[0]: the reference's CPU-runnable case, its protocol
                    result["cpu_baseline"] = cpu_baseline_protocol()
                except Exception as exc:  # never lose the GPU line to a host-side hiccup
                    result["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:  # second, labelled entry: the CPU oracle (checker port) on a slice of the headline workload
                port, sample = cpu_baseline(geo, args.batch, args.ctx, args.cpu_layers, args.quant)
                result["cpu_baseline"]["oracle_port"] = port
            except Exception as exc:
                sample = None
                result["cpu_baseline"]["oracle_port"] = {"error": f"{type(exc).__name__}: {exc}"}
            if sample is not None:
                try:  # the oracle's layer vs the HIP layer on the same inputs (outside the timed region)
                    pc = parity_check(geo, args.quant, sample)
                    if args.dtype == "bf16":
                        pc["what"] += " (fp16 weights / activations: the oracle restates the reference, which is fp16-only)"
                except Exception as exc:
                    pc = {"ok": False, "error": f"{type(exc).__name__}: {exc}"}
                result["parity_check"] = pc["ok"]
                result["parity_detail"] = pc
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        ps.destroy_parallel()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
