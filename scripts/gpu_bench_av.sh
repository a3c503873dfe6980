mkdir -p gpurun_out
python bench.py --workload avsr --steps 5 --warmup 2 > gpurun_out/bench_av.json 2> gpurun_out/bench_av.err; echo "rc=$?"; tail -3 gpurun_out/bench_av.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_av -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --workload avsr --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_av.log 2>&1
