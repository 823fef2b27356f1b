AB=$PWD/lite_llama_amd/lib/ab
for v in default head default head; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-14,70-200
done
for v in default head; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L timeout 300 python bench.py --no-cpu-baseline --steps 64 --warmup 8 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
