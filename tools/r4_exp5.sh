O=gpurun_out/r4e5; mkdir -p $O
AB=$PWD/lite_llama_amd/lib/ab
for v in abl4 abl20; do
for ap in 0 1; do
  LL_LIB_OVERRIDE=$AB/$v.so LL_GEMM3_AP=$ap PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/$v ap$ap /" | tee -a $O/ab.txt
done; done
