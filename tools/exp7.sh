O=gpurun_out/exp7; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
for v in default prio1 prio3 memonly_prio3; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/prio.txt
done
