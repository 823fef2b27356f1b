O=gpurun_out/r4e11; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "w4a16 or prepacked" 2>&1 | tail -3 | tee $O/pytest.txt
for v in default nopf default nopf default nopf; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L ONLY=gateup PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/ab.txt
done
for oc in -1 5 6; do LL_GEMM3_OC=$oc ONLY=gateup PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | tee -a $O/ab.txt; done
LL_GEMM3_OC=4 OCN=108 LL_LIB_OVERRIDE=$AB/tl.so timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | grep -A12 "gate|up wave 0" | tee $O/timeline_oc4.txt
