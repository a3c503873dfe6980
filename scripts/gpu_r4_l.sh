#!/bin/bash
# wave reductions on the lane network (common.h wave_sum / wave_max): whole suite, smoke, A/B of the benches against the previous build
set -o pipefail
mkdir -p gpurun_out
OLD=$PWD/gpurun_in/lib_old/libtavsr_hip.so
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4l_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/r4l_pytest_gpu.log | tail -2
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export TAVSR_LIB=$OLD; else unset TAVSR_LIB; fi
  python bench.py --mode fwd-encoder 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1])['fwd_encoder']; print('$lib fwd', d['layers12_eval_graph']['ms'], d.get('layers12_train_graph', {}).get('ms'))"
  python bench.py --workload asr --no-cpu-baseline --no-box --no-roofline --no-fwd-encoder --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib asr', d['value'], d['ms_per_step'], d.get('eager', {}).get('value'))"
  python bench.py --no-cpu-baseline --no-box --no-roofline --no-fwd-encoder --no-asr --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib av', d['value'], d['ms_per_step'])"
done; done
