cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 500 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_graph -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager "$@" > $R/gpurun_out/prof_graph.log 2>&1
echo rc=$?; tail -1 $R/gpurun_out/prof_graph.log | cut -c1-200
cd $R && python profiles/timeline.py gpurun_out/prof_graph/bench_results.db 3
