O=gpurun_out/r4e6; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "w4a16 or prepacked or headline" 2>&1 | tail -5 | tee $O/pytest.txt
for ap in 0 3 0 3; do
  LL_GEMM3_AP=$ap PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap$ap /" | tee -a $O/ab.txt
done
