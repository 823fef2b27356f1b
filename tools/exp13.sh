O=gpurun_out/exp13; mkdir -p $O
for dt in f16 bf16; do
timeout 400 python bench.py --model qwen2.5-1.5b --quant none --dtype $dt --batch 32 --no-cpu-baseline --steps 48 > $O/cfg2_$dt.json 2> $O/cfg2_$dt.err; echo rc=$?; tail -2 $O/cfg2_$dt.err | cut -c1-300
python -c "import json; d=json.load(open('$O/cfg2_$dt.json')); print('$dt', d['value'], d['ms_per_step'], d['step_roofline']['frac_of_8TBps'], d['roofline'] and (d['roofline']['frac'], d['roofline']['avg_launch_us']))"
done
