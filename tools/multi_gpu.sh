#!/bin/bash
# For a node with >= 2 MI355X (the driver's 8-GPU lease; never run by the builder: gpurun boxes have one GPU).  One rank per
# device over RCCL (backend "nccl") -- the tests place rank r on device r when device_count() >= world (tests/_dist.py) and
# assert that the collectives are RCCL's and that the one-shot kernel's peer buffers sit on distinct devices; bench.py starts
# its own ranks (python bench.py --gpus N) and self-checks the one-shot kernel against RCCL before using it (--allreduce auto).
# Usage: bash tools/multi_gpu.sh [outdir]     -> one JSON line per N in <outdir>/bench_tpN.json, test logs next to them
O=${1:-gpurun_out/multi_gpu}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices: $N" | tee $O/devices.txt
if [ "$N" -lt 2 ]; then echo "fewer than two devices: the tests fall back to ranks sharing device 0 (gloo host channel)"; fi
timeout 1800 python -m pytest tests/test_oneshot_allreduce_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
for G in 1 2 4 8; do
  if [ "$G" -le "$N" ] || [ "$G" -eq 1 ]; then
    timeout 900 python bench.py --gpus $G --steps 64 --warmup 8 --allreduce auto --no-cpu-baseline --no-secondary > $O/bench_tp$G.json 2> $O/bench_tp$G.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_tp$G.json")); c = d["config"]
    print("N=$G", d["value"], "tok/s", d["ms_per_step"], "ms", c["parallelism"], c["allreduce"], c["collective_backend"], "graph", d["graph"])
except Exception as e:
    print("N=$G", "ERR", e)
PY
  fi
done
