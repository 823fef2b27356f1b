mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_model_step.py -m gpu -q -k "moe or config5 or router or qwen3" > gpurun_out/r06/pytest_moe.txt 2>&1; tail -8 gpurun_out/r06/pytest_moe.txt
for lib in 1 0; do
  if [ $lib = 1 ]; then export LL_MOE_LIBRARY_GATE=1; else unset LL_MOE_LIBRARY_GATE; fi
  echo "== LL_MOE_LIBRARY_GATE=$lib"
  timeout 600 python bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --steps 24 --warmup 4 --as-secondary --no-prefill 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'parity_check')}, d.get('parity_detail', {}).get('argmax_agree'), d.get('parity_detail', {}).get('router_ties'))"
done
ROOT=$PWD; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pk5 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o b -- python $ROOT/bench.py --model qwen3-30b-a3b --quant fp8 --batch 64 --steps 12 --warmup 3 --as-secondary --no-prefill --no-cpu-baseline > /dev/null 2>&1; cd $ROOT
python tools/rocpd.py stats /tmp/pk5/b_results.db --by-grid 2>&1 | grep -v "at::native" | head -40 > gpurun_out/r06/cfg5_kernel_stats_bygrid_v0.txt; head -30 gpurun_out/r06/cfg5_kernel_stats_bygrid_v0.txt
