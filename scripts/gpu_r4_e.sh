# the whole GPU suite once more on one queue (TAVSR_SINGLE_STREAM=1) and once with NaN-poisoned allocator blocks
mkdir -p gpurun_out
( TAVSR_SINGLE_STREAM=1 timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_streams.py --deselect tests/test_gpu_switches.py ) > gpurun_out/r4e_single_stream.log 2>&1; echo "single-stream rc=$?"; grep -n "passed\|failed" gpurun_out/r4e_single_stream.log | tail -2
( TAVSR_POISON=nan timeout 3000 python -m pytest tests -m gpu -q ) > gpurun_out/r4e_poison.log 2>&1; echo "poison rc=$?"; grep -n "passed\|failed" gpurun_out/r4e_poison.log | tail -2
