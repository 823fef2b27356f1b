for a in 1 3 4 8 12; do
echo "== ablate $a (1 no dequant+MFMA, 2 no operand reads, 4 no weight DMA, 8 no activation DMA)"; LL_GEMM4=3 ONLY=down LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/v4_tl_a$a.so timeout 300 python benchmarks/gemm4_timeline.py 2>/dev/null | grep -E "consumer q0 heavy|X loader 0|W loader 0|clock"
done
