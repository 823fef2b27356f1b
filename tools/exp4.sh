O=gpurun_out/exp4; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
for v in c3_128 c3_256 c3_384 c3_388; do
  LL_LIB_OVERRIDE=$AB/$v.so PADS=0 timeout 200 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/ablate.txt
done
