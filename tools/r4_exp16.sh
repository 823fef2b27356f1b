for kw in "--steps 20 --warmup 5" "--steps 20 --warmup 5"; do
LL_BENCH_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary $kw 2>&1 >/tmp/out.json | grep "^\[" ; python -c "
import json,sys; d=json.loads(open('/tmp/out.json').read()); print('$kw', d['value'], d['ms_per_step'])"
done
