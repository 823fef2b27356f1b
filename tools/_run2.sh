mkdir -p gpurun_out
(
echo "== full"; timeout 300 python benchmarks/gemm_short.py | tail -1
for v in a1 a2 a4 a8 a3 a7 a11; do echo "== ablate $v"; LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/ss_$v.so timeout 300 python benchmarks/gemm_short.py | tail -1; done
for v in ss_tl ss_tl_a4; do echo "== timeline $v"; LL_LIB_OVERRIDE=lite_llama_amd/lib/ab/$v.so SHAPES=o timeout 300 python benchmarks/gemm_short_timeline.py; done
for m in 1 17 33 64; do M=$m TIME=0 SHAPES=qkv,o,qkv_tp2,o_tp2,down_tp8,gateup_tp8,c5_qkv,c5_o,l3_qkv,l3_o timeout 300 python benchmarks/gemm_short.py | tail -1; done
) > gpurun_out/ss3.log 2>&1
