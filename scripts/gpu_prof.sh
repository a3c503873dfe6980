# eager kernel trace of the headline step -> gpurun_out/prof_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --no-asr --no-box --sustain-s 0 "$@" > $R/gpurun_out/prof_bench.log 2>&1
echo rc=$?
cd $R && python profiles/summarize_rocpd.py gpurun_out/prof/bench_results.db 4 > gpurun_out/prof_stats.txt; head -30 gpurun_out/prof_stats.txt | cut -c1-150
rm -rf gpurun_out/prof
