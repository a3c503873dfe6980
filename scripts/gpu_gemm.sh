mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q > gpurun_out/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gemm.log
tail -5 gpurun_out/pytest_gemm.log
timeout 900 python profiles/gemm_sweep.py $1 > gpurun_out/gemm_sweep.txt 2>&1; echo "sweep rc=$?"
