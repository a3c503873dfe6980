# round 4, call d: whole GPU suite + smoke + default bench after the C-side sequencers
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r4d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/r4d_pytest_gpu.log | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
( time timeout 900 python bench.py ) > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d_bench.json').read().strip().split('\n')[-1])
print('AV', d['value'], d['ms_per_step'], 'eager', d['eager']['value'], 'sustained', d['sustained']['value'])
print('fwd', d['fwd_encoder']['layers12_eval_graph'], 'asr', d['asr']['value'], d['asr']['eager'], 'box', d['box']['fp32_mfma_tflops'], d['box']['hbm_copy_gb_per_s'])
PY
