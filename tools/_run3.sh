mkdir -p gpurun_out
(
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_model_step.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do
echo "== SS=0"; LL_GEMM_SS=0 timeout 300 python benchmarks/step_times.py | tail -1 | cut -c1-400
echo "== SS=1"; timeout 300 python benchmarks/step_times.py | tail -1 | cut -c1-400
done
) > gpurun_out/ss4.log 2>&1
