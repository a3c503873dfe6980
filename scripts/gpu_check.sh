# quick regression + timing check: parity tests, audio-only / AV bench lines, forward-encoder bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_av.py tests/test_gpu_ops.py tests/test_interctc.py -m gpu -x -q -k "not batch_32 and not 3200" 2>&1 | tail -2
for w in asr avsr; do timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', j['value'], j['ms_per_step'])"; done
timeout 600 python bench.py --mode fwd-encoder --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print({k:v['ms'] for k,v in d.items() if isinstance(v,dict) and 'graph' in k})"
