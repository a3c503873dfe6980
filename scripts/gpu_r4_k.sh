#!/bin/bash
# final matrix of the round: whole GPU suite (default, single-stream, race probes), smoke, default bench
set -o pipefail
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r4k_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/r4k_pytest_gpu.log | tail -2
( TAVSR_SINGLE_STREAM=1 timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4k_single_stream.log 2>&1; echo "single-stream rc=$?"; grep -n "passed\|failed" gpurun_out/r4k_single_stream.log | tail -2
bash scripts/gpu_probe_suite.sh
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
( time timeout 900 python bench.py ) > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4k_bench.json').read().strip().split('\n')[-1])
print('AV', d['value'], d['ms_per_step'], 'eager', d['eager']['value'], 'sustained', d['sustained']['value'])
print('fwd', d['fwd_encoder']['layers12_eval_graph'], 'asr', d['asr']['value'], d['asr']['eager'], 'box', d['box']['fp32_mfma_tflops'], d['box']['fp32_mfma_tflops_data'], d['box']['hbm_copy_gb_per_s'])
PY
