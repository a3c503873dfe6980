# round-end routine: records (bench lines, kernel tables, counters, decode), the whole GPU suite, smoke(), the race-amplifier suites
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
bash scripts/gpu_round_end.sh > gpurun_out/round_end.log 2>&1; echo "round end rc=$?"
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 ) > gpurun_out/suite_final.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/suite_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
bash scripts/gpu_probe_suite.sh > gpurun_out/probe_suite.log 2>&1; echo "probe suite rc=$?"; grep -E "passed|failed|smoke ok|==" gpurun_out/probe_suite.log | tail -8
timeout 300 python scripts/decode_stress.py 10 > gpurun_out/stress_final.log 2>&1; echo "stress rc=$?"; tail -1 gpurun_out/stress_final.log
