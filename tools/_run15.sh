mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "router or route" > gpurun_out/r06/pytest_moe.txt 2>&1; tail -4 gpurun_out/r06/pytest_moe.txt
ROOT=$PWD; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pk5 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o b -- python $ROOT/bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --steps 12 --warmup 3 --as-secondary --no-prefill --no-cpu-baseline > /tmp/pk5.json 2>&1; cd $ROOT
tail -c 600 /tmp/pk5.json
python tools/rocpd.py stats /tmp/pk5/b_results.db --by-grid 2>&1 | grep -v "at::native" | head -40 > gpurun_out/r06/cfg5_kernel_stats_bygrid_v0.txt; grep -E "moe_route|moe_gate|moe_align|Cijk" gpurun_out/r06/cfg5_kernel_stats_bygrid_v0.txt
