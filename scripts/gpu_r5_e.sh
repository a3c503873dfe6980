mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_beam_search.py -q -m gpu -x > gpurun_out/gemm_tests.log 2>&1; echo "gemm+beam tests rc=$?"; tail -5 gpurun_out/gemm_tests.log
timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1.json 2> gpurun_out/decode_b1.err; echo "decode1 rc=$?"; cut -c1-420 gpurun_out/decode_b1.json
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64.json 2> gpurun_out/decode_b64.err; echo "decode64 rc=$?"; cut -c1-420 gpurun_out/decode_b64.json
TAVSR_DECODE_LN_EPILOGUE=0 timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64_noln.json 2> gpurun_out/decode_b64_noln.err; echo "decode64 (separate LayerNorm launches) rc=$?"; cut -c1-420 gpurun_out/decode_b64_noln.json
TAVSR_DECODE_CTC_BESIDE=0 timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64_nospec.json 2> gpurun_out/decode_b64_nospec.err; echo "decode64 (CTC on the chain) rc=$?"; cut -c1-420 gpurun_out/decode_b64_nospec.json
timeout 600 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline > gpurun_out/decode_b256.json 2> gpurun_out/decode_b256.err; echo "decode256 rc=$?"; cut -c1-420 gpurun_out/decode_b256.json
bash scripts/gpu_decode_prof.sh 64 | head -24
