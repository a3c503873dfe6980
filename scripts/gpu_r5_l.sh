mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_beam_search.py -q -m gpu -x -k "laned or reused or captured" > gpurun_out/beam_tests.log 2>&1; echo "beam tests rc=$?"; tail -3 gpurun_out/beam_tests.log
for lanes in 1 2 4; do
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline --lanes $lanes > gpurun_out/d64_l$lanes.json 2>gpurun_out/d64_l$lanes.err; echo "lanes $lanes rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/d64_l$lanes.json'));print('batch 64, lanes $lanes:', d['utterances_per_s'], 'utt/s, search', d['search_s'], 'enc', d['encoder_s'], 'rtf p50', d['value'], 'tokens', d['tokens_decoded'])"
done
timeout 600 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline --lanes 4 > gpurun_out/d256_l4.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/d256_l4.json'));print('batch 256, lanes 4:', d['utterances_per_s'], 'utt/s, search', d['search_s'], 'enc', d['encoder_s'], 'tokens', d['tokens_decoded'])"
