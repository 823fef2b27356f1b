for v in none A B C D E; do
  echo "== variant $v"
  if [ $v = none ]; then unset LL_LIB_OVERRIDE; else export LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/v4_$v.so; fi
  LL_GEMM4=3 SHAPES=down timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1 | cut -c1-260
done
