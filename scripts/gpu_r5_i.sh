mkdir -p gpurun_out
for rq in 1 0 1 0; do
TAVSR_DECODE_RECORD_QUEUE=$rq timeout 600 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1_rq$rq.json 2> gpurun_out/decode_b1.err; echo "decode1 record queue=$rq rc=$?"; cut -c1-100 gpurun_out/decode_b1_rq$rq.json; python -c "
import json;d=json.load(open('gpurun_out/decode_b1_rq$rq.json'));print(d['value'], d['search_s'], d['tokens_decoded'], 1e6*d['search_s']/d['tokens_decoded']*91.4/91.4)"
done
timeout 600 python scripts/decode_chain_probe.py > gpurun_out/decode_chain_probe.txt 2>&1; echo "chain probe rc=$?"; grep "us per token" gpurun_out/decode_chain_probe.txt
