# implicit Conv3d stem: parity, then the AV bench line with the gather loader on / off
timeout 900 python -m pytest tests/test_gpu_stem.py tests/test_gpu_av.py tests/test_gpu_vsr.py -m gpu -x -q -k "not batch_32 and not 3200" 2>&1 | tail -40
for v in 1 0; do
  TAVSR_STEM_IMPLICIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('implicit=$v', j['value'], j['ms_per_step'], j['hbm_peak_gb'])"
done
