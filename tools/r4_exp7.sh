O=gpurun_out/r4e7; mkdir -p $O
timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "norm or partial" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 200 python benchmarks/norm_partials.py 2>&1 | tail -1 | tee -a $O/norm.txt
LL_GEMM3_XCD=4 LL_GEMM3_FILL=40 timeout 200 python benchmarks/norm_partials.py 2>&1 | tail -1 | tee -a $O/norm.txt
timeout 200 python benchmarks/norm_partials.py 2>&1 | tail -1 | tee -a $O/norm.txt
