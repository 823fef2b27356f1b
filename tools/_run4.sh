mkdir -p gpurun_out
(
for rs in "4 7" "4 8" "2 4" "8 8" "4 4" "2 7"; do set -- $rs; echo "== o R=$1 S=$2"; LL_GEMM_SS_R=$1 LL_GEMM_SS_S=$2 SHAPES=o timeout 200 python benchmarks/gemm_short.py | tail -1 | cut -c1-300; done
for rs in "4 7" "8 8" "4 6" "4 8" "2 7"; do set -- $rs; echo "== qkv R=$1 S=$2"; LL_GEMM_SS_R=$1 LL_GEMM_SS_S=$2 SHAPES=qkv timeout 200 python benchmarks/gemm_short.py | tail -1 | cut -c1-300; done
echo "== M=32 default"; M=32 SHAPES=qkv,o,l3_qkv,l3_o,c5_qkv,c5_o timeout 300 python benchmarks/gemm_short.py | tail -1
echo "== M=64 others"; SHAPES=l3_qkv,l3_o,c5_qkv,c5_o,qkv_tp2,o_tp2,down_tp8,gateup_tp8 timeout 300 python benchmarks/gemm_short.py | tail -1
) > gpurun_out/ss5.log 2>&1
