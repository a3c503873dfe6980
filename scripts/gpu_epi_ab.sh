mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_epi.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_epi.log
for w in asr avsr; do for f in 1 0 1 0; do
  TAVSR_GEMM_VEC_EPI=$f timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager > gpurun_out/epi.json 2> gpurun_out/epi.err; echo "$w vec_epi=$f rc=$? $(python -c "import json;j=json.loads(open('gpurun_out/epi.json').read().strip().splitlines()[-1]);print(j['value'], j['ms_per_step'])")"
done; done
for f in 1 0; do TAVSR_GEMM_VEC_EPI=$f timeout 600 python bench.py --mode fwd-encoder --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['fwd_encoder']; print('vec_epi=$f', {k:v['ms'] for k,v in d.items() if isinstance(v,dict) and 'graph' in k})"; done
TAVSR_GEMM_VEC_EPI=1 timeout 600 python profiles/gemm_shapes.py --workload asr --fwd-only 2>&1 | head -12
