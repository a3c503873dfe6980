# Conv3d stem routes: parity, then the AV bench line per route (padded 16-byte chunks / 4-byte gathers / im2col)
timeout 900 python -m pytest tests/test_gpu_stem.py tests/test_gpu_av.py tests/test_gpu_vsr.py -m gpu -x -q -k "not batch_32 and not 3200" 2>&1 | tail -12
for v in "TAVSR_X=0" "TAVSR_STEM_PAD16=0" "TAVSR_STEM_IMPLICIT=0" "TAVSR_X=0" "TAVSR_STEM_PAD16=0"; do
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], j['hbm_peak_gb'])"
done
