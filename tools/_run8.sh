mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "w8 or smoothquant" > gpurun_out/r06/pytest_w8.txt 2>&1; tail -15 gpurun_out/r06/pytest_w8.txt
timeout 300 python benchmarks/prefill_gemm8.py > gpurun_out/r06/prefill_gemm8.json 2>gpurun_out/r06/prefill_gemm8.err; tail -3 gpurun_out/r06/prefill_gemm8.json
LL_W8_NO_MTILED=1 timeout 600 python benchmarks/prefill_gemm8.py > gpurun_out/r06/prefill_gemm8_loop.json 2>>gpurun_out/r06/prefill_gemm8.err; tail -3 gpurun_out/r06/prefill_gemm8_loop.json
timeout 600 python bench.py --model llama-3-8b --quant smoothquant --batch 32 --steps 16 --warmup 4 --as-secondary > gpurun_out/r06/cfg4_line.json 2>gpurun_out/r06/cfg4_line.err; tail -c 1500 gpurun_out/r06/cfg4_line.json
