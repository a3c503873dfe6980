mkdir -p gpurun_out
for n in 2 4; do
  TAVSR_DP_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 5 --warmup 2 --sustain-s 3 --no-roofline --no-cpu-baseline --no-fwd-encoder > gpurun_out/dp_gloo$n.json 2> gpurun_out/dp_gloo$n.err
  echo "gloo x$n rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/dp_gloo$n.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('grad_exchange_exposed_ms_per_step'), d.get('sustained'), d['config']['grad_exchange'])"
done
TAVSR_DP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --sustain-s 0 --no-roofline --no-cpu-baseline --no-fwd-encoder --no-graph > gpurun_out/dp_gloo2_eager.json 2> gpurun_out/dp_gloo2_eager.err; echo "gloo x2 eager rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/dp_gloo2_eager.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('grad_exchange_exposed_ms_per_step'), d['config']['launch'])"
# the N > 1 step as two hipGraphs (early buckets exchanged under the second) against one graph: per-phase host times
bash scripts/gpu_split_rehearsal.sh
