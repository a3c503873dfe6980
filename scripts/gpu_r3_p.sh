timeout 600 python -m pytest tests/test_gpu_lin2.py -x -q 2>&1 | tail -5
timeout 300 python scripts/lin2_bench.py 2>&1 | grep -v amdgpu
