mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn.py -x -q > gpurun_out/pytest_attn.log 2>&1; echo "attn rc=$?"; tail -15 gpurun_out/pytest_attn.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_av.py tests/test_gpu_dropout.py tests/test_interctc.py tests/test_beam_search.py -m gpu -x -q > gpurun_out/pytest_par.log 2>&1; echo "parity rc=$?"; tail -8 gpurun_out/pytest_par.log
timeout 600 python bench.py --mode fwd-encoder --steps 20 --warmup 5 > gpurun_out/fwd_encoder.json 2> gpurun_out/fwd_encoder.err; echo "fwd rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/fwd_encoder.json'))['fwd_encoder']
for k,v in d.items():
    if isinstance(v,dict): print(k, v)"
for f in 1 0; do TAVSR_ATTN_FUSED=$f timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager > gpurun_out/asr_f$f.json 2> gpurun_out/asr_f$f.err; echo "asr fused=$f rc=$?"; cut -c1-200 gpurun_out/asr_f$f.json; done
