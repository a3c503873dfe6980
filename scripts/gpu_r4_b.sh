# round 4, call b: stream-safety suite, fused merge tail parity, layer / parity suites, forward target
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_streams.py -q --durations=10 ) > gpurun_out/r4b_streams.log 2>&1; echo "streams rc=$?"; tail -25 gpurun_out/r4b_streams.log
( time timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layer_c.py tests/test_gpu_parity.py tests/test_gpu_dropout.py -q -m gpu --durations=5 ) > gpurun_out/r4b_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r4b_tests.log
timeout 600 python bench.py --mode fwd-encoder > gpurun_out/r4b_fwd.json 2> gpurun_out/r4b_fwd.err; echo "fwd rc=$?"; cat gpurun_out/r4b_fwd.json
TAVSR_MERGE_PROJ=0 timeout 600 python bench.py --mode fwd-encoder > gpurun_out/r4b_fwd_off.json 2> gpurun_out/r4b_fwd_off.err; echo "fwd(off) rc=$?"; cat gpurun_out/r4b_fwd_off.json
