O=gpurun_out/r4e15; mkdir -p $O
timeout 900 python -m pytest tests/test_w4a16_prepacked_gpu.py -x -q -m gpu -k "unpack or compacted" 2>&1 | tail -5 | tee $O/pytest.txt
for kw in "" "--keep-reference-weights" ""; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $kw 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['memory']; print('$kw', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], m['weights_resident_bytes'], m['allocated_after_warmup_bytes'])"
done
