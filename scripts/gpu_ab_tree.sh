# A/B of two source trees in one box: bash scripts/gpu_ab_tree.sh DIR_A DIR_B [bench args]
a=$1; b=$2; shift; shift
mkdir -p gpurun_out
for rep in 1 2; do
for t in $a $b; do
  ( cd $t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" ) > gpurun_out/ab_tree.json 2> gpurun_out/ab_tree.err; echo "$t rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_tree.json").read().strip().splitlines()[-1])
    print("  ", j["value"], j["unit"], j["ms_per_step"], "ms/step")
except Exception as e:
    print("  no result:", e)
PY
done
done
