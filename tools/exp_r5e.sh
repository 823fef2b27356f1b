#!/bin/bash
O=$PWD/gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
for V in "" abl16 abl32; do
  if [ -z "$V" ]; then LIB=""; else LIB=$PWD/lite_llama_amd/lib/ab/v4_$V.so; fi
  LL_LIB_OVERRIDE=$LIB ONLY=gateup PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/xl_$V.json
  echo "variant=[$V] $(cat $O/xl_$V.json)"
done
for V in tl tl_abl1 tl_abl12 tl_abl16 tl_abl32; do
  echo "#### $V"
  ONLY="gate|up" LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/v4_$V.so timeout 200 python benchmarks/gemm4_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline_$V.txt | cut -c1-420
done
