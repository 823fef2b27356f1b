#!/bin/bash
timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_model_step.py -x -q -m gpu -k "mtiled or prefill_then_decode" 2>&1 | tail -3
for V in "" ""; do
  echo "[$V] $(timeout 300 python benchmarks/prefill_gemm.py 2>&1 | tail -1)"
done
