#!/bin/bash
O=$PWD/gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-prefill --steps 16 --warmup 3 > $O/bench_traced.json 2>/dev/null; cd $ROOT
python tools/rocpd.py stats /tmp/pk/b_results.db --by-grid 2>&1 | grep -v "at::native\|rocclr\|pack_" | head -16 | cut -c1-170
python tools/rocpd.py steps /tmp/pk/b_results.db 2>&1 | head -12
