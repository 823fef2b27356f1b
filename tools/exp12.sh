O=gpurun_out/exp12; mkdir -p $O
timeout 900 python bench.py --gpus 2 --steps 16 --warmup 4 > $O/bench_g2.json 2> $O/bench_g2.err; echo rc=$?; tail -2 $O/bench_g2.err | cut -c1-200; cut -c1-900 $O/bench_g2.json
LL_TP_NO_FUSED_NORM=1 timeout 900 python bench.py --gpus 2 --steps 16 --warmup 4 > $O/bench_g2_unfused.json 2> $O/bench_g2_unfused.err; echo rc=$?; cut -c1-300 $O/bench_g2_unfused.json
