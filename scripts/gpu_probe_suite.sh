# the parity suites, the beam search and smoke() with the race amplifier armed from the environment:
# every forked queue delayed at its head ("body"), then every owner delayed behind each join ("join")
mkdir -p gpurun_out
for mode in body join; do
  echo "== TAVSR_RACE_PROBE=200 TAVSR_RACE_PROBE_MODE=$mode"
  ( TAVSR_RACE_PROBE=200 TAVSR_RACE_PROBE_MODE=$mode timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_av.py tests/test_beam_search.py tests/test_gpu_layer_c.py tests/test_gpu_dropout.py -m gpu -q -x ) > gpurun_out/probe_suite_$mode.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" gpurun_out/probe_suite_$mode.log | tail -2
  TAVSR_RACE_PROBE=200 TAVSR_RACE_PROBE_MODE=$mode timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
done
