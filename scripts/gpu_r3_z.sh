python -m pytest tests/test_gpu_layer_c.py -x -q 2>&1 | tail -8
python -m pytest tests/test_gpu_parity.py tests/test_interctc.py -x -q 2>&1 | tail -3
for r in 1 2; do
for v in 1 0; do
  TAVSR_LAYER_C=$v timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr layer_c=$v graph', d['value'], 'eager', d['eager'])"
done
done
TAVSR_LAYER_C=1 python scripts/eager_host_profile.py 2>&1 | grep enqueue
TAVSR_LAYER_C=0 python scripts/eager_host_profile.py 2>&1 | grep enqueue
