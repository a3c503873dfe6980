python -m pytest tests/test_gpu_ffn2.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for r in 1 2; do
for v in 1 0; do
  TAVSR_FFN_BWD_SLAB=$v timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr slab=$v', d['value'], d['ms_per_step'])"
done
done
