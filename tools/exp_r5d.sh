#!/bin/bash
O=$PWD/gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python benchmarks/gemm4_check.py > $O/check.json 2> $O/check.err; head -c 200 $O/check.json; echo
EXTRA=1 LL_GEMM4_MINFILL=1 timeout 300 python benchmarks/gemm4_check.py > $O/check_extra.json 2> $O/check_extra.err; head -c 200 $O/check_extra.json; echo
LL_GEMM4=0 PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1
for V in "" r55 r65 r84 nowarm; do
  if [ -z "$V" ]; then LIB=""; else LIB=$PWD/lite_llama_amd/lib/ab/v4_$V.so; fi
  LL_LIB_OVERRIDE=$LIB PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/xl_$V.json
  echo "variant=[$V] $(cat $O/xl_$V.json)"
done
LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/v4_tl.so timeout 200 python benchmarks/gemm4_timeline.py 2>&1 | grep -v amdgpu.ids > $O/timeline.txt
cat $O/timeline.txt
for G in 0 1 3; do
  LL_GEMM4=$G STEPS=40 timeout 300 python benchmarks/step_times.py > $O/step_g$G.json 2> $O/step_g$G.err
  echo "G=$G $(cat $O/step_g$G.json | python -c 'import json,sys; d=json.load(sys.stdin); print(d["median_ms"])')"
done
