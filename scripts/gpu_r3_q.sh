for r in 1 2; do
for v in "0 512" "1 512" "1 100000"; do
  set -- $v
  TAVSR_LIN2=$1 TAVSR_LIN2_MIN_N=$2 timeout 600 python bench.py --mode fwd-encoder --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print('lin2=$1 min_n=$2', d['layers12_eval_graph'], d['layers12_train_graph'])"
done
done
