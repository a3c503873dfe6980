mkdir -p gpurun_out
for t in 1000 256 512 2000; do
  TAVSR_SPLIT_TARGET=$t timeout 600 python bench_decode.py --utterances 256 --batch 64 --no-cpu-baseline > gpurun_out/decode_t$t.json 2> gpurun_out/decode_t$t.err; echo "target $t rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/decode_t$t.json'));print(d['value'],d['utterances_per_s'],d['search_s'],d['encoder_s'])"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_decode -o dec -- python3 $R/bench_decode.py --utterances 64 --batch 64 --no-cpu-baseline > $R/gpurun_out/prof_decode.log 2>&1
cd $R && python profiles/summarize_rocpd.py gpurun_out/prof_decode/dec_results.db 1 > gpurun_out/decode_kernels.txt; head -24 gpurun_out/decode_kernels.txt | cut -c1-150
