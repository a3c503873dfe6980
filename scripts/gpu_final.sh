# round-end verification: smoke, full GPU suite, default bench record
mkdir -p gpurun_out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -4 gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
