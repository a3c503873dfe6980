mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_av.py -x -q > gpurun_out/pytest_av.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_av.log
tail -30 gpurun_out/pytest_av.log
