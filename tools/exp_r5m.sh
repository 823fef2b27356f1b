#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_model_step.py -x -q -m gpu -k "smoothquant or config4 or w8a8" 2>&1 | tail -8
for V in "X=1" "LL_W8A8_ROWS_OFF=1"; do
  echo "[$V] $(env $V python bench.py --model llama-3-8b --quant smoothquant --batch 32 --no-cpu-baseline --no-secondary --no-prefill --steps 32 2>/dev/null | python -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])')"
done
