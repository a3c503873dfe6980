for r in 1 2; do
for cfg in "1 1" "0 1" "0 0" "1 0"; do
  set -- $cfg
  TAVSR_AV_JOINT_FFN=$1 TAVSR_FFN2_BWD=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('av joint=$1 bwd=$2', d['value'], d['ms_per_step'])"
done
done
