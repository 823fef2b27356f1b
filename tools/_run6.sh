mkdir -p gpurun_out
(timeout 900 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary-configs --no-prefill > gpurun_out/bench_sim.json 2> gpurun_out/bench_sim.err; tail -3 gpurun_out/bench_sim.err) 
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
