O=gpurun_out/exp9; mkdir -p $O
for d in 5 4 3 2; do LL_GEMM3_GTDIV=$d PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/GTDIV=$d /" | cut -c1-40,95-200; done
for d in 5 3; do
  LL_GEMM3_GTDIV=$d timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GTDIV=$d', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
