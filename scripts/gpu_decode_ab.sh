# beam-search tests, then the decode bench (captured scorer step) and its eager-launch A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_beam_search.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench_decode.py --utterances 256 --batch 64 > gpurun_out/decode_g1.json 2> gpurun_out/decode_g1.err; echo rc=$?
cut -c1-520 gpurun_out/decode_g1.json
TAVSR_DECODE_GRAPH=0 timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_g0.json 2> gpurun_out/decode_g0.err; echo rc=$?
cut -c1-520 gpurun_out/decode_g0.json
timeout 600 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline > gpurun_out/decode_g1_b256.json 2>> gpurun_out/decode_g1.err; cut -c1-520 gpurun_out/decode_g1_b256.json
