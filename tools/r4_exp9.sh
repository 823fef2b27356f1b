O=gpurun_out/r4e9; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "beyond_eight or headline" 2>&1 | tail -5 | tee $O/pytest.txt
