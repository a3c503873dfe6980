python -m pytest tests/test_gpu_av.py tests/test_gpu_gemm.py tests/test_gpu_ops.py -x -q -k "conv or stage or resnet or dw or wgrad or frontend" 2>&1 | tail -3
bash scripts/gpu_ab_lib.sh "lib_b/libtavsr_hip.so tailored-avsr_amd/tavsr/lib/libtavsr_hip.so" --steps 10 --warmup 3 --sustain-s 0
