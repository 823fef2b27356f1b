O=gpurun_out/exp5; mkdir -p $O
for d in 5 4 3; do
  LL_GEMM3_GTDIV=$d timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GTDIV=$d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/gtdiv.txt
done
