# the whole GPU suite + smoke, as the driver runs them at round end
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
