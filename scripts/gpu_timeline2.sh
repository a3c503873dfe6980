# graph-replay kernel tables of two trees in one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for t in . _old; do
  tag=$(echo $t | tr -d './_'); tag=${tag:-cur}
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/$t/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_$tag.log 2>&1
  echo "$t rc=$?"
  cd $R && python profiles/timeline.py gpurun_out/prof_$tag/bench_results.db 3 --kernels > gpurun_out/timeline_$tag.txt
  tail -3 gpurun_out/timeline_$tag.txt
done
