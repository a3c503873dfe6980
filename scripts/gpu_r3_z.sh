python -m pytest tests/test_gpu_parity.py tests/test_interctc.py tests/test_train_harness.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_av.py -x -q -k "model or batch32 or golden" 2>&1 | tail -3
for r in 1 2; do
for v in 1 0; do
  TAVSR_LOSS_BRANCH=$v timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr loss_branch=$v', d['value'], d['ms_per_step'])"
done
done
for v in 1 0; do
  TAVSR_LOSS_BRANCH=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('av loss_branch=$v', d['value'], d['ms_per_step'])"
done
