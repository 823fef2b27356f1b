AB=$PWD/lite_llama_amd/lib/ab
for v in default rw7 rw5 rw4 default; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-14,70-200
done
