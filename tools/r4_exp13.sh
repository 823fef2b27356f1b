O=gpurun_out/r4e13; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4e13/bench.json"))
print(d["value"], d["ms_per_step"], d["step_roofline"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("parity_check"))
print(d.get("memory")); print(d.get("secondary")); print(d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"))
PY
