# kernel trace of the beam search at batch $1 (default 1) -> gpurun_out/decode_timeline_b$1.txt
B=${1:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 500 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_dec_b$B -o dec -- python3 $R/bench_decode.py --utterances $((B*2)) --batch $B --no-cpu-baseline > $R/gpurun_out/prof_dec_b$B.log 2>&1
echo rc=$?
cd $R && python profiles/decode_timeline.py gpurun_out/prof_dec_b$B/dec_results.db > gpurun_out/decode_timeline_b$B.txt; cut -c1-170 gpurun_out/decode_timeline_b$B.txt
python profiles/decode_timeline.py gpurun_out/prof_dec_b$B/dec_results.db --sequence > gpurun_out/decode_sequence_b$B.txt
rm -rf gpurun_out/prof_dec_b$B
