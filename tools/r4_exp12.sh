O=gpurun_out/r4e12; mkdir -p $O
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py -x -q -m gpu -k "unpack or compacted" 2>&1 | tail -30 | tee $O/pytest.txt
