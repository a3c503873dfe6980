#!/bin/bash
# host side of the un-captured step: one allocation per layer for the kept state / the gradients, descriptor templates
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_layer_c.py tests/test_gpu_streams.py tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_dropout.py tests/test_dp_gloo.py -x -q -m gpu 2>&1 | tail -3
python scripts/eager_host_profile.py 2>&1 | grep -v amdgpu.ids | head -16 | cut -c1-170
for i in 1 2; do
python bench.py --workload asr --no-cpu-baseline --no-box --no-roofline --no-fwd-encoder --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('asr', d['value'], d['ms_per_step'], d.get('eager'))"
done
python -m pytest tests/test_gpu_ops.py tests/test_data_pipeline.py tests/test_gpu_av.py -x -q -m gpu -k "cut_to_longest or pipeline or model" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-box --no-roofline --no-fwd-encoder --no-asr --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('av', d['value'], d['ms_per_step'], d.get('eager'))"
