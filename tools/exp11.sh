O=gpurun_out/exp11; mkdir -p $O
timeout 1500 python bench.py --gpus 8 --steps 8 --warmup 2 > $O/bench_g8.json 2> $O/bench_g8.err; echo rc=$?; tail -3 $O/bench_g8.err | cut -c1-300; cat $O/bench_g8.json | cut -c1-1400
