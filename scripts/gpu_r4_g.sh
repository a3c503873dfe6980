#!/bin/bash
# activations on the hardware transcendentals (common.h): micro A/B against the previous build (gpurun_in/lib_old), whole GPU suite,
# smoke, the benches of both builds in one call
set -o pipefail
mkdir -p gpurun_out
OLD=$PWD/gpurun_in/lib_old/libtavsr_hip.so
for lib in old new; do
  echo "== $lib"
  if [ $lib = old ]; then export TAVSR_LIB=$OLD; else unset TAVSR_LIB; fi
  python scripts/gelu_cost.py 2>&1 | grep -v amdgpu.ids
  python scripts/act_bwd_bench.py 2>&1 | grep -v amdgpu.ids
done
unset TAVSR_LIB
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r4g_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/r4g_pytest_gpu.log | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export TAVSR_LIB=$OLD; else unset TAVSR_LIB; fi
  python bench.py --mode fwd-encoder 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['fwd_encoder']; print('$lib fwd', d['layers12_eval_graph'], d.get('layers12_train_graph'))"
  python bench.py --workload asr --no-cpu-baseline --no-box --no-roofline --no-fwd-encoder --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib asr', d['value'], d['ms_per_step'], d.get('eager', {}).get('value'))"
done; done
unset TAVSR_LIB
( time timeout 900 python bench.py ) > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4g_bench.json').read().strip().split('\n')[-1])
print('AV', d['value'], d['ms_per_step'], 'eager', d['eager']['value'], 'sustained', d['sustained']['value'])
print('fwd', d['fwd_encoder']['layers12_eval_graph'], 'asr', d['asr']['value'], d['asr']['eager'], 'box', d['box']['fp32_mfma_tflops'], d['box']['fp32_mfma_tflops_data'], d['box']['hbm_copy_gb_per_s'])
PY
