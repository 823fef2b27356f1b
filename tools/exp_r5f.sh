#!/bin/bash
O=$PWD/gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json")); print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("prefill"), indent=1))
PY
tail -5 $O/bench.err
cd /tmp && rm -rf /tmp/pf && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1; cd $OLDPWD
python tools/rocpd.py stats /tmp/pf/p_results.db --by-grid 2>&1 | head -30 > $O/prefill_trace.txt; cat $O/prefill_trace.txt | cut -c1-200
