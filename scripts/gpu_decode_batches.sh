mkdir -p gpurun_out
for b in 1 8 32 64 256; do
  n=$((b*4)); if [ $n -lt 16 ]; then n=16; fi
  timeout 600 python bench_decode.py --utterances $n --batch $b --no-cpu-baseline > gpurun_out/decode_b$b.json 2> gpurun_out/decode_b$b.err
  python -c "import json;d=json.load(open('gpurun_out/decode_b$b.json'));print('batch',$b,'p50 RTF',d['value'],'utt/s',d['utterances_per_s'],'enc_s',d['encoder_s'],'search_s',d['search_s'],'n',d['utterances'])"
done
