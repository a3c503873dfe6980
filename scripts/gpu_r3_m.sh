for r in 1 2; do
for v in 0 1; do
  TAVSR_ATTN_STAGED=$v timeout 600 python bench.py --mode fwd-encoder --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print('staged=$v', d['layers12_eval_graph'], d['layers12_train_graph'])"
done
done
for v in 0 1; do
  TAVSR_ATTN_STAGED=$v timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('asr staged=$v', d['value'], d['ms_per_step'])"
done
