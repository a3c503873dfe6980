# usage: [OUT=name] bash scripts/gpu_pmc_hbm.sh [bench.py flags, e.g. --workload asr]
# HBM byte counters of the benchmark step: one rocprofv3 pass per counter (both together exceed the hardware's
# counter budget), --pmc with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --no-asr --no-box --sustain-s 0 "$@" > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"; grep -v "^    @" $R/gpurun_out/pmc_$c.log | tail -2 | cut -c1-300
done
cd $R
python - <<'PY'
import shutil
with open("gpurun_out/pmc_hbm_counters.csv", "w") as out:
    first = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            with open(f"gpurun_out/pmc_{c}/p_counter_collection.csv") as f:
                head = f.readline()
                if first:
                    out.write(head)
                    first = False
                shutil.copyfileobj(f, out)
        except FileNotFoundError as e:
            print("missing", e)
PY
OUT=${OUT:-pmc_hbm}
python profiles/summarize_pmc.py gpurun_out/pmc_hbm_counters.csv 2 gpurun_out/$OUT.json > gpurun_out/$OUT.txt; head -14 gpurun_out/$OUT.txt
rm -rf gpurun_out/pmc_hbm_counters.csv gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
