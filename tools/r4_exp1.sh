# round 4, experiment 1: owner / contributor split of the fused gate|up launch -- parity, A/B against stream-K, timeline
O=gpurun_out/r4e1; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "w4a16 or prepacked or headline" 2>&1 | tail -3 | tee $O/pytest.txt
for oc in -1 4 2 3 -1 4; do
  LL_GEMM3_OC=$oc ONLY=gateup PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | tee -a $O/ab.txt
done
LL_GEMM3_OC=4 OCN=108 LL_LIB_OVERRIDE=$AB/tl.so timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | grep -A12 "gate|up wave" | tee $O/timeline_oc4.txt
LL_GEMM3_OC=-1 LL_LIB_OVERRIDE=$AB/tl.so timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | grep -A10 "gate|up wave" | tee $O/timeline_streamk.txt
