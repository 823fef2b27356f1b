O=gpurun_out/r4tests; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
