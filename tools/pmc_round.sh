#!/bin/bash
# MFMA / VALU / LDS utilisation counters of the final tree (north_star: "MFMA-utilisation counters"): two SQ passes + FETCH / WRITE
# passes over the decode step's eager launches, for the headline (int4) and for config 4 (int8 MFMA).  Run on the GPU box:
#   bash tools/pmc_round.sh r06 v1        -> gpurun_out/r06/pmc_sq_*_v1.{txt,md}, pmc_{FETCH,WRITE}_SIZE_*_v1.txt, pmc_traffic_v1.json
R=${1:-r06}; V=${2:-v1}; O=$PWD/gpurun_out/$R; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp
A="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run_pass() {  # tag, counters, bench flags...
  local tag=$1; local ctrs=$2; shift 2
  cd /tmp && rm -rf /tmp/pmc_$tag && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$tag -o p -- python $ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --no-prefill --steps 3 --warmup 1 "$@" > /dev/null 2> $O/pmc_$tag.err; cd $ROOT
}
run_pass hA "$A"
run_pass hB "$B"
rm -f $O/pmc_sq_$V.json
PMC_SQ_JSON=$O/pmc_sq_$V.json python tools/pmc_sq.py /tmp/pmc_hA/p_results.db /tmp/pmc_hB/p_results.db wgemm4 wgemm3 wss_kernel fd_stage1 skip_rmsnorm_partials > $O/pmc_sq_headline_$V.md 2>&1
python tools/rocpd.py pmc /tmp/pmc_hA/p_results.db wgemm > $O/pmc_sq_headline_passA_$V.txt 2>&1
python tools/rocpd.py pmc /tmp/pmc_hA/p_results.db wss_kernel >> $O/pmc_sq_headline_passA_$V.txt 2>&1
python tools/rocpd.py pmc /tmp/pmc_hB/p_results.db wgemm > $O/pmc_sq_headline_passB_$V.txt 2>&1
python tools/rocpd.py pmc /tmp/pmc_hB/p_results.db wss_kernel >> $O/pmc_sq_headline_passB_$V.txt 2>&1
run_pass cA "$A" --model llama-3-8b --quant smoothquant --batch 32
run_pass cB "$B" --model llama-3-8b --quant smoothquant --batch 32
PMC_SQ_JSON=$O/pmc_sq_$V.json python tools/pmc_sq.py /tmp/pmc_cA/p_results.db /tmp/pmc_cB/p_results.db dense8_kernel wgemm16_rows skip_rmsnorm_q8 w8a8 fd_stage1 > $O/pmc_sq_cfg4_$V.md 2>&1
# prefill GEMM (M-tiled): MFMA utilisation of the 64 x 512 prompt pass
cd /tmp && rm -rf /tmp/pmc_pA /tmp/pmc_pB
timeout 600 rocprofv3 --kernel-trace --pmc $A -d /tmp/pmc_pA -o p -- python $ROOT/benchmarks/prefill_gemm.py > /dev/null 2> $O/pmc_pA.err
timeout 600 rocprofv3 --kernel-trace --pmc $B -d /tmp/pmc_pB -o p -- python $ROOT/benchmarks/prefill_gemm.py > /dev/null 2> $O/pmc_pB.err
cd $ROOT
PMC_SQ_JSON=$O/pmc_sq_$V.json python tools/pmc_sq.py /tmp/pmc_pA/p_results.db /tmp/pmc_pB/p_results.db wgemm_prefill > $O/pmc_sq_prefill_$V.md 2>&1
# the 8-bit M-tiled prefill engine (round 6): int8 x int8 / fp8 -> fp16 MFMA utilisation
cd /tmp && rm -rf /tmp/pmc_qA /tmp/pmc_qB
timeout 600 rocprofv3 --kernel-trace --pmc $A -d /tmp/pmc_qA -o p -- python $ROOT/benchmarks/prefill_gemm8.py > /dev/null 2> $O/pmc_qA.err
timeout 600 rocprofv3 --kernel-trace --pmc $B -d /tmp/pmc_qB -o p -- python $ROOT/benchmarks/prefill_gemm8.py > /dev/null 2> $O/pmc_qB.err
cd $ROOT
PMC_SQ_JSON=$O/pmc_sq_$V.json python tools/pmc_sq.py /tmp/pmc_qA/p_results.db /tmp/pmc_qB/p_results.db w8_mtiled >> $O/pmc_sq_prefill_$V.md 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  run_pass $C $C
  python tools/rocpd.py pmc /tmp/pmc_$C/p_results.db wgemm > $O/pmc_${C}_gemm_$V.txt 2>&1
  python tools/rocpd.py pmc /tmp/pmc_$C/p_results.db wss_kernel >> $O/pmc_${C}_gemm_$V.txt 2>&1
  python tools/rocpd.py pmc /tmp/pmc_$C/p_results.db fd_stage1 > $O/pmc_${C}_attention_$V.txt 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/p_results.db /tmp/pmc_WRITE_SIZE/p_results.db ${R}_$V > $O/pmc_traffic_$V.json 2>&1
head -8 $O/pmc_sq_headline_$V.md; head -8 $O/pmc_sq_cfg4_$V.md; head -4 $O/pmc_sq_prefill_$V.md; head -8 $O/pmc_traffic_$V.json
