# A/B inside one call: the layers' weight gradients on the chain's own queue (0) or beside it (1), enqueued at the end of the layer's own
# backward ("end") or carried to the next layer's ("ffn": behind its first feed-forward block, "join": behind its branch join)
mkdir -p gpurun_out
W=${1:-asr}
for rep in 1 2; do
for cfg in "0 4" "1 4"; do
  set -- $cfg
  TAVSR_WGRAD_BESIDE=$1 TAVSR_DEC_WGRAD=$2 timeout 600 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-decode --no-asr --no-fwd-encoder --no-box --sustain-s 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -5 gpurun_out/ab.err
  python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"beside={sys.argv[1]} at={sys.argv[2]:5s}: {d['value']:8.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  sustained {d.get('sustained', {}).get('value')}  eager {d.get('eager', {}).get('value')}  hbm_peak {d.get('hbm_peak_gb')} GB")
PY
done; done
