mkdir -p gpurun_out
timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/d64.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/d64.json'));print('one process, batch 64:', d['utterances_per_s'], 'utt/s, search', d['search_s'], 'enc', d['encoder_s'])"
timeout 600 python bench_decode.py --utterances 256 --batch 32 --no-cpu-baseline > gpurun_out/d32.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/d32.json'));print('one process, batch 32:', d['utterances_per_s'], 'utt/s, search', d['search_s'], 'enc', d['encoder_s'])"
timeout 600 python bench_decode.py --utterances 256 --batch 32 --no-cpu-baseline > gpurun_out/d32a.json 2>/dev/null &
timeout 600 python bench_decode.py --utterances 256 --batch 32 --no-cpu-baseline > gpurun_out/d32b.json 2>/dev/null &
wait
python -c "
import json
a=json.load(open('gpurun_out/d32a.json'));b=json.load(open('gpurun_out/d32b.json'))
print('two processes at once, batch 32 each:', a['utterances_per_s'], '+', b['utterances_per_s'], '=', a['utterances_per_s']+b['utterances_per_s'], 'utt/s (walls', a['wall_s'], b['wall_s'], ')')"
