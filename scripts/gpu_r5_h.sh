mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_rowlin.py tests/test_beam_search.py -q -m gpu -x > gpurun_out/beam_tests.log 2>&1; echo "rowlin + beam tests rc=$?"; tail -3 gpurun_out/beam_tests.log
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_b64.json 2> gpurun_out/decode_b64.err; echo "decode64 rc=$?"; cut -c1-470 gpurun_out/decode_b64.json
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline --pipeline > gpurun_out/decode_b64_pipe.json 2> gpurun_out/decode_b64_pipe.err; echo "decode64 pipelined rc=$?"; cut -c1-470 gpurun_out/decode_b64_pipe.json
timeout 600 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline --pipeline > gpurun_out/decode_b256_pipe.json 2> gpurun_out/decode_b256_pipe.err; echo "decode256 pipelined rc=$?"; cut -c1-470 gpurun_out/decode_b256_pipe.json
timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1.json 2> gpurun_out/decode_b1.err; echo "decode1 rc=$?"; cut -c1-420 gpurun_out/decode_b1.json
timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline --pipeline > gpurun_out/decode_b1_pipe.json 2> gpurun_out/decode_b1_pipe.err; echo "decode1 pipelined rc=$?"; cut -c1-420 gpurun_out/decode_b1_pipe.json
