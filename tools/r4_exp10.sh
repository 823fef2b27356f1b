O=gpurun_out/r4e10; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
for v in default sf1 sf0 default sf1 sf0; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/ab.txt
done
