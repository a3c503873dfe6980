# A/B of library builds in one box: bash scripts/gpu_ab_lib.sh "path1 path2 ..." [bench args]
libs=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for l in $libs; do
  TAVSR_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager "$@" > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err; echo "$l rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
    print("  ", j["value"], j["unit"], j["ms_per_step"], "ms/step")
except Exception as e:
    print("  no result:", e)
PY
done
done
