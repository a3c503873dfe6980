#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ffn2.py -x -q 2>&1 | tail -5 | tee gpurun_out/r3e_ffn2_tests.txt
python scripts/ffn2_trace.py 10,4 10,3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e_trace.txt
timeout 600 python scripts/ffn2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e_ffn2_bench.txt
