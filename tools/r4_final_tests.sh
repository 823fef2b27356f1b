O=gpurun_out/r4final; mkdir -p $O
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
