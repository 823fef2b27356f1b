for pad in 0 64 192 8; do
  echo "== XPAD=$pad"
  XPAD=$pad SHAPES=qkv,o,down timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1
  XPAD=$pad FULL=1 SHAPES=gateup timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1
done
