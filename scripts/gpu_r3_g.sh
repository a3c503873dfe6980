#!/bin/bash
# round 3: AV parity with the joint feed-forward call, DP rehearsal (gloo ranks sharing one GPU), default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_av.py tests/test_gpu_parity.py tests/test_gpu_dropout.py tests/test_dp_gloo.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r3g_tests.txt
for n in 2 4; do
  TAVSR_DP_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 5 --warmup 2 --sustain-s 0 --no-roofline > gpurun_out/r3g_dp_gloo$n.json 2> gpurun_out/r3g_dp_gloo$n.err
  echo "gloo x$n rc=$?"; tail -c 1500 gpurun_out/r3g_dp_gloo$n.json; tail -3 gpurun_out/r3g_dp_gloo$n.err
done
timeout 1200 python bench.py > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3g_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "sustained", "eager") if k in d})
print({k: v for k, v in d.get("fwd_encoder", {}).items() if k.startswith("layers12")})
print(d.get("roofline", {}).get("frac"), d.get("cpu_baseline", {}).get("value"))
PY
