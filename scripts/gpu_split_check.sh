#!/bin/bash
# two-graph step (tavsr.dp.TwoPhaseBackward): parity test, 1-rank overhead A/B, gloo world-2 rehearsal with and without the split
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_av.py -x -q -m gpu -k "two_graphs or graph_replayed" 2>&1 | tail -3
F="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0"
for r in 1 2; do for v in "" "--split-backward"; do for wl in avsr asr; do
  x=$(timeout 600 python bench.py $F --workload $wl $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'][:40])")
  echo "round $r $wl [$v] $x"
done; done; done | tee gpurun_out/split_ab.txt
for v in "" "--no-split-backward"; do
  TAVSR_DP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --sustain-s 3 --no-roofline --no-cpu-baseline --no-fwd-encoder $v > gpurun_out/dp_split.json 2> gpurun_out/dp_split.err
  echo "gloo x2 [$v] rc=$?"; grep "two-graph" gpurun_out/dp_split.err | tail -1; python -c "
import json; d=json.loads(open('gpurun_out/dp_split.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('grad_exchange_exposed_ms_per_step'), d.get('sustained'), d['config']['launch'])"
done 2>&1 | tee gpurun_out/split_rehearsal.txt
