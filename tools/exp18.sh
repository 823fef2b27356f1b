AB=$PWD/lite_llama_amd/lib/ab
for v in default m4 m4_32 m4_384 m4_32_384 m4_64; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-14,70-200
done
