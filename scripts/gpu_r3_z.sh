for r in 1 2 3; do
for v in 1 0; do
  TAVSR_LAYER_C=$v timeout 600 python bench.py --mode fwd-encoder --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print('layer_c=$v eval', d['layers12_eval_graph']['ms'], 'train', d['layers12_train_graph']['ms'], 'eager eval', d['layers12_eval_eager']['ms'], 'eager train', d['layers12_train_eager']['ms'])"
done
done
