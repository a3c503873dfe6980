# implicit second convolution of Conv2dSubsampling: parity, then the audio-only bench line with the route on / off
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_av.py -m gpu -x -q -k "not batch_32 and not 3200" 2>&1 | tail -3
for rep in 1 2; do for v in "TAVSR_CONV2_IMPLICIT=1" "TAVSR_CONV2_IMPLICIT=0"; do
  env $v timeout 600 python bench.py --workload asr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], j['hbm_peak_gb'])"
done; done
