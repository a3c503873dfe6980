#!/bin/bash
# gloo world-2 rehearsal (both ranks on the one GPU) of the N > 1 step: two hipGraphs with early buckets leaving under the second,
# against one hipGraph with every bucket after it; per-phase host times on stderr (TAVSR_BENCH_TRACE)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in "" "--no-split-backward"; do
  tag=$([ -z "$v" ] && echo split || echo whole)
  TAVSR_BENCH_TRACE=1 TAVSR_DP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 4 --warmup 2 --sustain-s 0 --no-roofline --no-cpu-baseline --no-fwd-encoder $v > gpurun_out/dp_gloo2_$tag.json 2> gpurun_out/dp_gloo2_$tag.err
  echo "gloo x2 [$tag] rc=$?"; grep "two-graph\|rank 0\] step" gpurun_out/dp_gloo2_$tag.err | tail -8; python -c "
import json; d=json.loads(open('gpurun_out/dp_gloo2_$tag.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('grad_exchange_exposed_ms_per_step'), d['config']['launch'])"
done 2>&1 | tee gpurun_out/split_rehearsal.txt
