# generic in-call A/B of the AV bench line: bash scripts/gpu_ab.sh "VAR=a" "VAR=b" ...   (each given twice, alternating)
for rep in 1 2; do for v in "$@"; do
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'])"
done; done
