# A/B inside ONE gpurun call (boxes of the pool differ by 1 - 2 %): bench.py under each environment given, twice, alternating.
# usage: bash scripts/ab.sh asr "TAVSR_WGRAD_BESIDE=0" "TAVSR_WGRAD_BESIDE=1" ...
mkdir -p gpurun_out
W=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-decode --no-asr --no-fwd-encoder --no-box --sustain-s 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -5 gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s}: {d['value']:8.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  sustained {d.get('sustained', {}).get('value')}  eager {d.get('eager', {}).get('value')}  hbm_peak {d.get('hbm_peak_gb')} GB")
PY
done; done
