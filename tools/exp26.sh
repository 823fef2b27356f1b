AB=$PWD/lite_llama_amd/lib/ab
LL_LIB_OVERRIDE=$AB/tl.so SHAPES=4608x3584 PARTIALS=1 NORMED=1 timeout 300 python benchmarks/gemm3_timeline.py 2>&1 | grep -v amdgpu.ids
