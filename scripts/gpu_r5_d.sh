mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_rowlin.py tests/test_beam_search.py -q -m gpu -x > gpurun_out/rowlin_tests.log 2>&1; echo "rowlin+beam tests rc=$?"; tail -3 gpurun_out/rowlin_tests.log
timeout 600 python scripts/decode_chain_probe.py > gpurun_out/decode_chain_probe.txt 2>&1; echo "chain probe rc=$?"; tail -14 gpurun_out/decode_chain_probe.txt
timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1.json 2> gpurun_out/decode_b1.err; echo "decode1 rc=$?"; cut -c1-420 gpurun_out/decode_b1.json
TAVSR_DECODE_CTC_BESIDE=0 timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1_nospec.json 2> gpurun_out/decode_b1_nospec.err; echo "decode1 (CTC on the chain) rc=$?"; cut -c1-420 gpurun_out/decode_b1_nospec.json
bash scripts/gpu_decode_prof.sh 1 | head -24
