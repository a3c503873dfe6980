#!/bin/bash
# in-call A/B of one env switch on both bench lines: bash scripts/gpu_ab2.sh "VAR=a" "VAR=b" ...  (interleaved, two rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
F="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0"
for r in 1 2; do
  for v in "$@"; do
    for wl in avsr asr; do
      x=$(env $v timeout 600 python bench.py $F --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
      echo "round $r $wl $v $x"
    done
  done
done | tee gpurun_out/ab2.txt
