#!/bin/bash
O=$PWD/gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_model_step.py -x -q -m gpu -k "moe or config5 or qwen3" 2>&1 | tail -8
for V in "" "LL_MOE_ALIGN_V1=1 LL_MOE_NO_INTERLEAVE=1"; do
  env $V timeout 600 python bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --no-cpu-baseline --no-secondary --no-prefill --steps 16 --warmup 4 > $O/cfg5_$(echo $V | tr -c 'A-Za-z0-9' '_').json 2>$O/cfg5.err
  python - <<PY
import json,glob
d=json.load(open("$O/cfg5_$(echo $V | tr -c 'A-Za-z0-9' '_').json")); print("[$V]", d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("avg_block_us"))
PY
done
tail -3 $O/cfg5.err
