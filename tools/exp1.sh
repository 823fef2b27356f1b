#!/bin/bash
# round-3 experiment batch 1 (GEMM memory side); run on the GPU box from the repo root
O=gpurun_out/exp1; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
run() { tag=$1; shift; env "$@" timeout 300 python benchmarks/gemm3_xlayout.py > $O/$tag.log 2>&1; tail -1 $O/$tag.log; }
run base PADS=0,64,128,192,256,512,1024,8
run nt0 LL_LIB_OVERRIDE=$AB/nt0.so PADS=0,64
run nt3 LL_LIB_OVERRIDE=$AB/nt3.so PADS=0,64
run xcm LL_GEMM3_XCM=1 PADS=0

run warm COPIES=1 PADS=0,64
run memonly LL_LIB_OVERRIDE=$AB/memonly.so PADS=0,64
run memonly_xcm LL_LIB_OVERRIDE=$AB/memonly.so LL_GEMM3_XCM=1 PADS=0

run componly LL_LIB_OVERRIDE=$AB/componly.so PADS=0
run base2 PADS=0
timeout 300 python bench.py --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; tail -c 600 $O/bench_base.json
