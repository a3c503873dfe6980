python -m pytest tests/test_gpu_ops.py -x -q -k "csgu or dwconv" 2>&1 | tail -3
python scripts/csgu_bench.py 2>&1 | grep -v amdgpu
for r in 1 2; do
for v in 0 1; do
  TAVSR_CSGU_FUSED=$v timeout 600 python bench.py --mode fwd-encoder --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print('csgu=$v', d['layers12_eval_graph'], d['layers12_train_graph'])"
done
done
for v in 0 1; do
  TAVSR_CSGU_FUSED=$v timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr csgu=$v', d['value'], d['ms_per_step'])"
  TAVSR_CSGU_FUSED=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('av csgu=$v', d['value'], d['ms_per_step'])"
done
