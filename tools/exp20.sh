AB=$PWD/lite_llama_amd/lib/ab
for m in 64 32 16; do
for v in default m4 componly; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L M=$m PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/M=$m $v /" | cut -c1-20,80-220
done; done
