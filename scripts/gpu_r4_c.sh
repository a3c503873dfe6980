# round 4, call c: after the prune - whole GPU suite (incl. streams + switches), smoke, forward target
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=10 ) > gpurun_out/r4c_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 gpurun_out/r4c_pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py --mode fwd-encoder 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['fwd_encoder']; print({k:v['ms'] for k,v in d.items() if isinstance(v,dict)})"
