for c in 1 0; do
  if [ $c = 1 ]; then export COPIES=1; else unset COPIES; fi
  echo "== COPIES=$c"
  SHAPES=qkv,o,down timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1
  FULL=1 SHAPES=gateup timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1
done
