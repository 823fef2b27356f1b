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
# Day-one checklist (round-4 review, item 9), BEFORE any bench: (1) peer access for every ordered pair of devices -- what the
# one-shot all-reduce's peer mappings need over xGMI; (2) per world size, the one-shot kernel's start-up check against RCCL on
# the step's payload, with the reason when it falls back (bench.py writes the same reason into config.allreduce_fallback_reason)
python - <<PY | tee $O/peer_access.txt
import torch
from lite_llama_amd.distributed.parallel_state import peer_access_matrix
m = peer_access_matrix()
print("hipDeviceCanAccessPeer (row = accessing device):")
for i, row in enumerate(m):
    print(f"  dev {i}: " + " ".join(str(v) for v in row))
print("all pairs peer-accessible:", all(all(r) for r in m))
PY
for G in 2 4 8; do
  if [ "$G" -le "$N" ]; then
    timeout 600 python bench.py --gpus $G --steps 4 --warmup 2 --allreduce auto --no-cpu-baseline --no-secondary --no-prefill > $O/check_tp$G.json 2> $O/check_tp$G.err
    python - <<PY | tee -a $O/peer_access.txt
import json
try:
    c = json.load(open("$O/check_tp$G.json"))["config"]
    print("world $G: collective =", c["allreduce"], "| fallback reason:", c.get("allreduce_fallback_reason"), "| devices:", c.get("rank_devices"))
except Exception as e:
    print("world $G: ERR", e, open("$O/check_tp$G.err").read()[-400:])
PY
  fi
done
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
