# round 4, first GPU call: the stream-safety suite, then the whole GPU suite + smoke, then the default bench line
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_streams.py -x -q --durations=10 ) > gpurun_out/r4a_streams.log 2>&1; echo "streams rc=$?"; tail -15 gpurun_out/r4a_streams.log
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 --deselect tests/test_gpu_streams.py ) > gpurun_out/r4a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4a_pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
( time timeout 900 python bench.py ) > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r4a_bench.json
