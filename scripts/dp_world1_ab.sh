# the N > 1 step's exchange path on ONE GPU (a one-rank RCCL communicator) against the plain step, inside one call:
# what the two-graph split and the communication queue's waits cost per step.   usage: bash scripts/dp_world1_ab.sh [asr|avsr]
mkdir -p gpurun_out
W=${1:-asr}
COMMON="--workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-decode --no-asr --no-fwd-encoder --no-box --no-eager --sustain-s 2"
for rep in 1 2; do
for cfg in "" "--split-backward" "--split-backward --force-rccl" "--force-rccl"; do
  timeout 600 python bench.py $COMMON $cfg > gpurun_out/dpab.json 2> gpurun_out/dpab.err || tail -5 gpurun_out/dpab.err
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/dpab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1] or '(one graph, no exchange)':36s}: {d['value']:8.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step   launch: {d['config'].get('launch')}   exchange: {d['config'].get('grad_exchange')}"[:260], flush=True)
PY
done; done
