O=gpurun_out/exp10; mkdir -p $O
echo nt1; CTX=512,584,640,1024,2048 timeout 200 python benchmarks/attn_ctx.py 2>&1 | grep ctx
echo nt0; LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/fdnt0.so CTX=512,584,640,1024,2048 timeout 200 python benchmarks/attn_ctx.py 2>&1 | grep ctx
timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench nt1', d['value'], d['ms_per_step'])"
LL_LIB_OVERRIDE=$PWD/lite_llama_amd/lib/ab/fdnt0.so timeout 300 python bench.py --no-cpu-baseline --steps 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench nt0', d['value'], d['ms_per_step'])"
