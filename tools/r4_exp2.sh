O=gpurun_out/r4e2; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
timeout 600 python -m pytest tests/test_w4a16_prepacked_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "sticky or route_topk or moe_align" 2>&1 | tail -3 | tee $O/pytest.txt
LL_GEMM3_OC=4 OCN=108 LL_LIB_OVERRIDE=$AB/tl.so timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | grep -A12 "gate|up wave" | tee $O/timeline_oc4.txt
