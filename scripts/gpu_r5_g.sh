mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_rowlin.py -q -m gpu -x > gpurun_out/rowlin_tests.log 2>&1; echo "rowlin tests rc=$?"; tail -3 gpurun_out/rowlin_tests.log
timeout 600 python scripts/tree_group_bench.py > gpurun_out/tree_group_bench.txt 2>&1; echo rc=$?; grep -E "^keys|distinct ancestor" gpurun_out/tree_group_bench.txt
timeout 300 python scripts/tree_attn_bench.py > gpurun_out/tree_attn_bench.txt 2>&1; echo "tree bench rc=$?"; grep "^LM" gpurun_out/tree_attn_bench.txt
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64.json 2> gpurun_out/decode_b64.err; echo "decode64 rc=$?"; cut -c1-420 gpurun_out/decode_b64.json
timeout 600 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline > gpurun_out/decode_b256.json 2> gpurun_out/decode_b256.err; echo "decode256 rc=$?"; cut -c1-420 gpurun_out/decode_b256.json
