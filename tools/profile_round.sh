#!/bin/bash
# Round profile artefacts (run on the GPU box from the repo root): the driver-protocol bench line (with its prefill object and the
# secondary configurations), the protocol-defaults line, kernel traces of the headline step, of the prefill pass and of configs 2 / 4 / 5.
# PMC passes: tools/pmc_round.sh.  Usage: bash tools/profile_round.sh r06 v1
R=${1:-r06}; V=${2:-v1}; O=$PWD/gpurun_out/$R; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$V.json 2> $O/bench_$V.err
cd /tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-prefill --steps 16 --warmup 3 > $O/bench_traced_$V.json 2>/dev/null; cd $ROOT
python tools/rocpd.py stats /tmp/pk/b_results.db --by-grid > $O/bench_kernel_stats_bygrid_$V.txt 2>&1
python tools/rocpd.py steps /tmp/pk/b_results.db >> $O/bench_kernel_stats_bygrid_$V.txt 2>&1
cd /tmp && rm -rf /tmp/pf && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o p -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1; cd $ROOT
python tools/rocpd.py stats /tmp/pf/p_results.db --by-grid 2>&1 | head -24 > $O/prefill_kernel_stats_bygrid_$V.txt
for C in "cfg2 --model qwen2.5-1.5b --quant none --dtype bf16 --batch 32" "cfg4 --model llama-3-8b --quant smoothquant --batch 32" "cfg5 --model qwen3-30b-a3b --quant fp8 --batch 64"; do
  set -- $C; N=$1; shift
  cd /tmp && rm -rf /tmp/pk_$N && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pk_$N -o c -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-secondary --no-prefill --steps 8 --warmup 2 > /dev/null 2>&1; cd $ROOT
  python tools/rocpd.py stats /tmp/pk_$N/c_results.db --by-grid 2>&1 | head -40 > $O/${N}_kernel_stats_bygrid_$V.txt
  python tools/rocpd.py steps /tmp/pk_$N/c_results.db 2>&1 | head -3 >> $O/${N}_kernel_stats_bygrid_$V.txt
done
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_$V.json") if l.startswith("{")][-1]); r=d["roofline"]
print("headline", d["value"], d["ms_per_step"], d["step_roofline"]["frac_of_8TBps"], r["frac"], r["avg_launch_us"], d.get("parity_check"), "ttft", d["prefill"]["ttft_ms"], d["prefill"]["achieved_TFLOPs"])
for e in d.get("secondary", []):
    print("  ", e.get("config", e.get("workload")), e.get("value"), e.get("ms_per_step"), e.get("step_roofline_frac_of_8TBps"), (e.get("roofline") or {}).get("frac"), e.get("parity_check"), e.get("error"))
print("drop-in route", (d.get("reference_order") or {}).get("ms_per_step"), "shard_sim", [(p["tp"], p["ms_per_step"]) for p in (d.get("shard_sim") or {}).get("points", [])])
PY
