#!/bin/bash
O=$PWD/gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
for V in "" abl1 abl2 abl12 abl8 abl4; do
  if [ -z "$V" ]; then LIB=""; else LIB=$PWD/lite_llama_amd/lib/ab/v4_$V.so; fi
  LL_LIB_OVERRIDE=$LIB ONLY=gateup,down PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/xl_$V.json
  echo "variant=[$V] $(cat $O/xl_$V.json)"
done
LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/v4_tl.so timeout 200 python benchmarks/gemm4_timeline.py 2>&1 | grep -v amdgpu.ids > $O/timeline.txt
cat $O/timeline.txt
