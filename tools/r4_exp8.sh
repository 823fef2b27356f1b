O=gpurun_out/r4e8; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
for v in default ps1 ps2 default ps1 ps2; do
  if [ $v = default ]; then L=""; else L="LL_LIB_OVERRIDE=$AB/$v.so"; fi
  env $L timeout 200 python benchmarks/norm_partials.py 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/norm.txt
done
