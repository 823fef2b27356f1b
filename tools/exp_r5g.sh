#!/bin/bash
O=$PWD/gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_model_step.py tests/test_kernels_gpu.py -x -q -m gpu -k "mtiled or prefill_then_decode or smoothquant_with_bias or swiglu or gate_up or activations" 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); print(d["value"], d["ms_per_step"]); p=d.get("prefill"); print(p["ttft_ms"], p["achieved_TFLOPs"], p["frac_of_mfma_peak"])
PY
tail -3 $O/bench.err
cd /tmp && rm -rf /tmp/pf && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1; cd $OLDPWD
python tools/rocpd.py stats /tmp/pf/p_results.db --by-grid 2>&1 | head -9 > $O/prefill_trace.txt; cat $O/prefill_trace.txt | cut -c1-180
