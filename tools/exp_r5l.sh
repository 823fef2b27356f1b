#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_model_step.py -x -q -m gpu -k "dense16 or row_group or swiglu or config2 or hip_model_matches or unquantised" 2>&1 | tail -8
for V in "X=1" "LL_DENSE16_ROWS_OFF=1"; do
  echo "[$V] $(env $V python bench.py --model qwen2.5-1.5b --quant none --dtype bf16 --batch 32 --no-cpu-baseline --no-secondary --no-prefill --steps 48 2>/dev/null | python -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])')"
  echo "[$V headline] $(env $V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-prefill 2>/dev/null | python -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"])')"
done
