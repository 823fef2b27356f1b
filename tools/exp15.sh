AB=$PWD/lite_llama_amd/lib/ab
timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "w4a16 or prepacked or engine_graph or headline" 2>&1 | tail -2
for v in default a_m1w0 a_m0w1 default; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-12,70-200
done
timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
