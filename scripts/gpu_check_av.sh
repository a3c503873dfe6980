mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_av.py -m gpu -x -q -k "not batch_32 and not 3200" 2>&1 | tail -2
for f in 1 0 1 0; do TAVSR_FRONT_PAIR=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$f', j['value'], j['ms_per_step'])"; done
