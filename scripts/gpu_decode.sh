# decode benchmark + kernel table of one batch
mkdir -p gpurun_out
timeout 900 python bench_decode.py --utterances 256 --batch 64 > gpurun_out/decode.json 2> gpurun_out/decode.err; echo rc=$?
cat gpurun_out/decode.json | cut -c1-700
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_decode -o dec -- python3 $R/bench_decode.py --utterances 64 --batch 64 --no-cpu-baseline > $R/gpurun_out/prof_decode.log 2>&1
cd $R && python profiles/summarize_rocpd.py gpurun_out/prof_decode/dec_results.db 1 > gpurun_out/decode_kernels.txt; head -28 gpurun_out/decode_kernels.txt | cut -c1-150
