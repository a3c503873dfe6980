python -m pytest tests/test_gpu_av.py -x -q -k "stage or resnet or conv" 2>&1 | tail -3
for r in 1 2; do
for v in 1 2 3; do
  TAVSR_CONV_TILE=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('av conv_tile=$v', d['value'], d['ms_per_step'])"
done
done
