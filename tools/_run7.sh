mkdir -p gpurun_out/r06
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-prefill --no-reference-order --shard-sim none --steps 16 --warmup 3 > $ROOT/gpurun_out/r06/bench_traced_v0.json 2>/dev/null; cd $ROOT
python tools/rocpd.py stats /tmp/pk/b_results.db --by-grid > gpurun_out/r06/bench_kernel_stats_bygrid_v0.txt 2>&1
python tools/rocpd.py steps /tmp/pk/b_results.db >> gpurun_out/r06/bench_kernel_stats_bygrid_v0.txt 2>&1
timeout 1500 python -m pytest tests/test_oneshot_allreduce_gpu.py tests/test_paged_kv_gpu.py tests/test_w4a16_prepacked_gpu.py tests/test_weights_loader_gpu.py tests/test_configs_gpu.py -m gpu -q > gpurun_out/pytest_gpu2.txt 2>&1; tail -5 gpurun_out/pytest_gpu2.txt
