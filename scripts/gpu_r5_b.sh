# round 5, call B: decode after the one-launch beam update / CTC beside the scorers / single-barrier LayerNorm / folded gathers
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_rowlin.py tests/test_beam_search.py -q -m gpu -x > gpurun_out/rowlin_tests.log 2>&1; echo "rowlin+beam tests rc=$?"; tail -8 gpurun_out/rowlin_tests.log
timeout 300 python scripts/rowlin_bench.py > gpurun_out/rowlin_bench.txt 2>&1; echo "rowlin bench rc=$?"; tail -9 gpurun_out/rowlin_bench.txt
timeout 600 python scripts/decode_chain_probe.py > gpurun_out/decode_chain_probe.txt 2>&1; echo "chain probe rc=$?"; tail -9 gpurun_out/decode_chain_probe.txt
timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1.json 2> gpurun_out/decode_b1.err; echo "decode1 rc=$?"; cut -c1-420 gpurun_out/decode_b1.json
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64.json 2> gpurun_out/decode_b64.err; echo "decode64 rc=$?"; cut -c1-420 gpurun_out/decode_b64.json
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline --cold-capture > gpurun_out/decode_b64_cold.json 2> gpurun_out/decode_b64_cold.err; echo "decode64 cold rc=$?"; cut -c1-420 gpurun_out/decode_b64_cold.json
( time timeout 1200 python -m pytest tests -q -m gpu -x -v --durations=15 ) > gpurun_out/suite_b.log 2>&1; echo suite rc=$?; grep -n "PASSED\|FAILED" gpurun_out/suite_b.log | tail -3; tail -25 gpurun_out/suite_b.log
