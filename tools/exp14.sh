O=gpurun_out/exp14; mkdir -p $O
timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "w4a16 or prepacked" 2>&1 | tail -2
PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
