python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_av.py -x -q 2>&1 | tail -3
for r in 1 2; do
  timeout 600 python bench.py --mode fwd-encoder --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print(d['layers12_eval_graph'], d['layers12_train_graph'])"
done
timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr', d['value'], d['ms_per_step'])"
