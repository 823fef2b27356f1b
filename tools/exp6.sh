O=gpurun_out/exp6; mkdir -p $O
timeout 600 python bench.py --steps 48 > $O/bench.json 2> $O/bench.err; echo rc=$?; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('parity_check'), d.get('parity_detail'), d.get('hbm'), d['roofline']['frac'])"
timeout 900 python bench.py --gpus 2 --steps 16 --warmup 4 > $O/bench_g2.json 2> $O/bench_g2.err; echo rc=$?; tail -5 $O/bench_g2.err; cat $O/bench_g2.json | cut -c1-1500
