# A/B of one environment switch on the headline bench: bash scripts/gpu_ab.sh VAR [extra bench args]
var=$1; shift
mkdir -p gpurun_out
for v in 1 0 1 0; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab_${var}_$v.json 2> gpurun_out/ab_${var}_$v.err; echo "$var=$v rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_${var}_$v.json").read().strip().splitlines()[-1])
    print("  ", j["value"], j["unit"], j["ms_per_step"], "ms/step")
except Exception as e:
    print("  no result:", e)
PY
done
