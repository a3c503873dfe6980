python -m pytest tests/test_gpu_layer_c.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --sustain-s 0 --no-graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr eager', d['value'], d['ms_per_step'], d['hbm_peak_gb'])"
