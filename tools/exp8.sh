O=gpurun_out/exp8; mkdir -p $O
PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1
PARTIALS=1 SHAPES_ONLY=o LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/tl.so python benchmarks/gemm3_timeline.py 2>&1 > $O/tl.log; grep -B0 -A4 "^== o \|^== down " $O/tl.log | cut -c1-250
