timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_model_step.py -x -q -m gpu -k "in_launch_norm" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --steps 64 --warmup 8 2>&1 | tail -1 | cut -c1-400
