mkdir -p gpurun_out
(
echo "== default"; SHAPES=qkv,o,qkv_tp2,o_tp2,down_tp8 timeout 300 python benchmarks/gemm_short.py | tail -1
echo "== M=32"; M=32 SHAPES=qkv,o timeout 300 python benchmarks/gemm_short.py | tail -1
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py -m gpu -x -q -k "short_stream or partial" 2>&1 | tail -3
for i in 1 2; do
echo "== SS=0"; LL_GEMM_SS=0 timeout 300 python benchmarks/step_times.py | tail -1 | cut -c230-400
echo "== SS=1"; timeout 300 python benchmarks/step_times.py | tail -1 | cut -c230-400
done
) > gpurun_out/ss6.log 2>&1
