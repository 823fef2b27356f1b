mkdir -p gpurun_out/r06
for cfg in "" "LL_GEMM4=3" "LL_GEMM4=3 LL_GEMM4_RGP=2"; do
  echo "== $cfg"; env $cfg SHAPES=down timeout 300 python benchmarks/gemm_short.py 2>/dev/null | tail -1
done
echo "== timeline v4 down (RGP 4)"; LL_GEMM4=3 ONLY=down LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/v4_tl.so timeout 300 python benchmarks/gemm4_timeline.py 2>/dev/null
echo "== timeline v4 down (RGP 2)"; LL_GEMM4=3 LL_GEMM4_RGP=2 ONLY=down LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/v4_tl.so timeout 300 python benchmarks/gemm4_timeline.py 2>/dev/null
