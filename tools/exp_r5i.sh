#!/bin/bash
O=$PWD/gpurun_out/r5i; mkdir -p $O
for rep in 1 2; do
for V in "" prio1 prio3; do
  if [ -z "$V" ]; then LIB=""; else LIB=$PWD/lite_llama_amd/lib/ab/v4_$V.so; fi
  LL_LIB_OVERRIDE=$LIB ONLY=gateup PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/xl_$V.json
  echo "variant=[$V] $(cat $O/xl_$V.json)"
done; done
