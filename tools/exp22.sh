timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "w4a16 or prepacked or headline" 2>&1 | tail -2
AB=$PWD/lite_llama_amd/lib/ab
for v in default nointer default nointer; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-14,70-200
done
