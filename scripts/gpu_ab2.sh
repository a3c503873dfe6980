# A/B of an environment variable over given values: bash scripts/gpu_ab2.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in $vals; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > gpurun_out/ab_${var}_$v.json 2> gpurun_out/ab_${var}_$v.err; echo "$var=$v rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_${var}_$v.json").read().strip().splitlines()[-1])
    print("  ", j["value"], j["unit"], j["ms_per_step"], "ms/step")
except Exception as e:
    print("  no result:", e)
PY
done
done
