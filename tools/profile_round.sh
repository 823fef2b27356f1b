#!/bin/bash
# Round profile artefacts (run on the GPU box from the repo root): bench line + kernel trace of the same command, PMC
# FETCH / WRITE passes over the step's eager launches, secondary configurations.  Usage: bash tools/profile_round.sh r04 v1
R=${1:-r04}; V=${2:-v1}; O=$PWD/gpurun_out/$R; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 --reference-order > $O/bench_$V.json 2> $O/bench_$V.err
timeout 900 python bench.py > $O/bench_defaults_$V.json 2> $O/bench_defaults_$V.err
cd /tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 16 --warmup 3 > $O/bench_traced_$V.json 2>/dev/null; cd $ROOT
python tools/rocpd.py stats /tmp/pk/b_results.db --by-grid > $O/bench_kernel_stats_bygrid_$V.txt 2>&1
python tools/rocpd.py stats /tmp/pk/b_results.db > $O/bench_kernel_stats_$V.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/pmc_$C && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o p -- python $ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > /dev/null 2>&1; cd $ROOT
  python tools/rocpd.py pmc /tmp/pmc_$C/p_results.db wgemm3 > $O/pmc_${C}_gemm_$V.txt 2>&1
  python tools/rocpd.py pmc /tmp/pmc_$C/p_results.db fd_stage1 > $O/pmc_${C}_attention_$V.txt 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/p_results.db /tmp/pmc_WRITE_SIZE/p_results.db ${R}_$V > $O/pmc_traffic_$V.json 2>&1
timeout 400 python bench.py --ctx 2048 --no-cpu-baseline --no-secondary --steps 32 > $O/bench_ctx2048_$V.json 2>/dev/null
timeout 400 python bench.py --model qwen2.5-1.5b --quant none --dtype bf16 --batch 32 --no-cpu-baseline --no-secondary --steps 48 > $O/bench_cfg2_qwen2.5-1.5b_bf16_b32_$V.json 2>/dev/null
timeout 400 python bench.py --model qwen2.5-1.5b --quant none --dtype f16 --batch 32 --no-cpu-baseline --no-secondary --steps 48 > $O/bench_cfg2_qwen2.5-1.5b_fp16_b32_$V.json 2>/dev/null
timeout 400 python bench.py --model llama-3-8b --quant smoothquant --batch 32 --no-cpu-baseline --no-secondary --steps 32 > $O/bench_cfg4_llama3-8b_w8a8_b32_$V.json 2>/dev/null
timeout 600 python bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --no-cpu-baseline --no-secondary --steps 16 --warmup 4 > $O/bench_cfg5_qwen3-30b-a3b_fp8_b64_$V.json 2>/dev/null
cd /tmp && rm -rf /tmp/pk2 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o c -- python $ROOT/bench.py --model qwen2.5-1.5b --quant none --dtype bf16 --batch 32 --no-cpu-baseline --no-secondary --steps 16 --warmup 3 > /dev/null 2>&1; cd $ROOT
python tools/rocpd.py stats /tmp/pk2/c_results.db --by-grid 2>&1 | head -40 > $O/cfg2_kernel_stats_bygrid_$V.txt
cd /tmp && rm -rf /tmp/pk4 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk4 -o c -- python $ROOT/bench.py --model llama-3-8b --quant smoothquant --batch 32 --no-cpu-baseline --no-secondary --steps 8 --warmup 2 > /dev/null 2>&1; cd $ROOT
python tools/rocpd.py stats /tmp/pk4/c_results.db --by-grid 2>&1 | head -40 > $O/cfg4_kernel_stats_bygrid_$V.txt
python tools/rocpd.py steps /tmp/pk4/c_results.db 2>&1 | head -3 >> $O/cfg4_kernel_stats_bygrid_$V.txt
cd /tmp && rm -rf /tmp/pk5 && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o c -- python $ROOT/bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --no-cpu-baseline --no-secondary --steps 8 --warmup 2 > /dev/null 2>&1; cd $ROOT
python tools/rocpd.py stats /tmp/pk5/c_results.db --by-grid 2>&1 | head -40 > $O/cfg5_kernel_stats_bygrid_$V.txt
python tools/rocpd.py steps /tmp/pk5/c_results.db 2>&1 | head -3 >> $O/cfg5_kernel_stats_bygrid_$V.txt
ls -la $O | tail -20
python - <<PY
import json
for f in ["bench_$V.json","bench_traced_$V.json","bench_ctx2048_$V.json","bench_cfg2_qwen2.5-1.5b_bf16_b32_$V.json","bench_cfg2_qwen2.5-1.5b_fp16_b32_$V.json","bench_cfg4_llama3-8b_w8a8_b32_$V.json","bench_cfg5_qwen3-30b-a3b_fp8_b64_$V.json"]:
    try:
        d=json.load(open("$O/"+f)); r=d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], d["step_roofline"]["frac_of_8TBps"], r.get("frac"), r.get("avg_launch_us"), d.get("parity_check"), (d.get("reference_order") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
