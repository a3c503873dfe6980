# round 5, call A: the two-graph test that crashed the suite (alone; then with the old bypass walk), launch floor, rowlin
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_av.py -q -m gpu -k "two_graphs" > gpurun_out/two_graphs.log 2>&1; echo "two_graphs alone rc=$?"; tail -5 gpurun_out/two_graphs.log
timeout 600 python - > gpurun_out/two_graphs_nowalk.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tailored-avsr_amd"); sys.path.insert(0, "tests")
from tavsr import dp
dp.TwoPhaseBackward.bypassed = lambda self, loss: False
import pytest
raise SystemExit(pytest.main(["tests/test_gpu_av.py", "-q", "-m", "gpu", "-k", "two_graphs"]))
PY
echo "two_graphs without the walk rc=$?"; tail -3 gpurun_out/two_graphs_nowalk.log
timeout 300 python scripts/launch_floor.py > gpurun_out/launch_floor.txt 2>&1; echo "floor rc=$?"; cat gpurun_out/launch_floor.txt
timeout 900 python -m pytest tests/test_gpu_rowlin.py tests/test_beam_search.py -q -m gpu > gpurun_out/rowlin_tests.log 2>&1; echo "rowlin+beam tests rc=$?"; tail -8 gpurun_out/rowlin_tests.log
timeout 300 python scripts/rowlin_bench.py > gpurun_out/rowlin_bench.txt 2>&1; echo "rowlin bench rc=$?"; tail -12 gpurun_out/rowlin_bench.txt
timeout 600 python scripts/decode_chain_probe.py > gpurun_out/decode_chain_probe.txt 2>&1; echo "chain probe rc=$?"; cat gpurun_out/decode_chain_probe.txt | tail -10
