#!/bin/bash
# round 5, GPU call A: row-group engine correctness + same-box A/B against the unit-loop engine
O=$PWD/gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python benchmarks/gemm4_check.py > $O/check.json 2> $O/check.err
head -c 600 $O/check.json; tail -3 $O/check.err
EXTRA=1 LL_GEMM4_MINFILL=1 timeout 300 python benchmarks/gemm4_check.py > $O/check_extra.json 2> $O/check_extra.err
head -c 600 $O/check_extra.json; tail -3 $O/check_extra.err
for G in 0 3; do
  LL_GEMM4=$G PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py > $O/xlayout_g$G.txt 2>&1
  tail -5 $O/xlayout_g$G.txt
done
for G in 0 1 2 3; do
  LL_GEMM4=$G STEPS=40 timeout 300 python benchmarks/step_times.py > $O/step_g$G.json 2> $O/step_g$G.err
  echo "G=$G $(cat $O/step_g$G.json | head -c 400)"
done
