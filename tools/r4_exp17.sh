timeout 300 python benchmarks/step_times.py 2>/dev/null | tail -1
IDLE=3 timeout 300 python benchmarks/step_times.py 2>/dev/null | tail -1
