AB=$PWD/lite_llama_amd/lib/ab
for v in default ps1 ps2 default; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-12,70-200
done
for v in default ps1 ps2; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench $v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
