# in-launch add-and-normalise: parity, then the step with / without it (same box)
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py -x -q -m gpu -k "in_launch_norm" 2>&1 | tail -4
for v in on off on off; do
  E=""
  [ $v = off ] && E="LL_NO_NORM_IN_GEMM=1"
  env $E timeout 300 python bench.py --no-cpu-baseline --steps 64 --warmup 8 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('parity_check'))"
done
