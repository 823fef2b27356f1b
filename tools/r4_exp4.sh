O=gpurun_out/r4e4; mkdir -p $O
for ap in 0 1 0 1; do
  LL_GEMM3_AP=$ap ONLY=qkv,o,down PADS=0 timeout 300 python benchmarks/gemm3_xlayout.py 2>&1 | tail -1 | sed "s/^/ap$ap /" | tee -a $O/ab.txt
done
