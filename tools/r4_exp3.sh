O=gpurun_out/r4e3; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py tests/test_model_step.py -x -q -m gpu -k "w4a16 or prepacked or headline or swiglu" 2>&1 | tail -5 | tee $O/pytest.txt
for ap in 0 1 0 1; do
  LL_GEMM3_AP=$ap PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap$ap /" | tee -a $O/ab.txt
done
LL_GEMM3_AP=1 LL_GEMM3_NF=1 ONLY=gateup PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap1nf1 /" | tee -a $O/ab.txt
LL_GEMM3_AP=1 LL_GEMM3_NF=1 LL_GEMM3_OC=2 ONLY=gateup PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap1nf1oc2 /" | tee -a $O/ab.txt
LL_GEMM3_AP=1 LL_GEMM3_NF=1 LL_GEMM3_OC=-1 ONLY=gateup PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap1nf1streamk /" | tee -a $O/ab.txt
PARTIALS=1 LL_GEMM3_AP=1 LL_LIB_OVERRIDE=$AB/tl.so timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | tee $O/timeline_ap1.txt | grep -A8 "down wave\|qkv wave 0" | head -40
