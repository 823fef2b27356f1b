O=gpurun_out/r4e14; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
for kw in "" "--keep-reference-weights"; do
cd /tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace -d /tmp/pk -o b -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 16 --warmup 3 $kw 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$kw', d['value'], d['ms_per_step'], d['memory'])"; cd $ROOT
echo "== $kw"; python tools/rocpd_steps.py /tmp/pk/b_results.db
done
