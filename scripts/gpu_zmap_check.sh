# slice-per-XCD mapping of the convolution weight gradients: parity, then the AV bench line with the mapping on / off
timeout 1200 python -m pytest tests/test_gpu_stem.py tests/test_gpu_av.py tests/test_gpu_gemm.py -m gpu -x -q -k "not batch_32" > gpurun_out/zmap_tests.log 2>&1; tail -3 gpurun_out/zmap_tests.log
for v in 1 0 1 0; do
  TAVSR_CONV_ZMAP=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zmap=$v', j['value'], j['ms_per_step'])"
done
