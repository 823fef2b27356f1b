timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py -x -q -m gpu -k "in_launch_norm" 2>&1 | tail -2
timeout 300 python benchmarks/norm_in_gemm.py 2>&1 | tail -1
timeout 300 python benchmarks/norm_in_gemm.py 2>&1 | tail -1
