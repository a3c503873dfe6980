mkdir -p gpurun_out
run() {
  env "$@" timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_x.json 2> gpurun_out/decode_x.err; echo "$* rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/decode_x.json'));print(d['value'],d['utterances_per_s'],d['search_s'],d['encoder_s'])"
}
run TAVSR_BRANCH_STREAM=1
run TAVSR_BRANCH_STREAM=0
run TAVSR_BRANCH_STREAM=1
run TAVSR_BRANCH_STREAM=0
