# round-end routine: records (bench lines, kernel tables, counters, decode), the whole GPU suite, smoke(), the race-amplifier suites
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
bash scripts/gpu_round_end.sh > gpurun_out/round_end.log 2>&1; echo "round end rc=$?"
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 ) > gpurun_out/suite_final.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/suite_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
bash scripts/gpu_probe_suite.sh > gpurun_out/probe_suite.log 2>&1; echo "probe suite rc=$?"; grep -E "passed|failed|smoke ok|==" gpurun_out/probe_suite.log | tail -8
timeout 300 python scripts/decode_stress.py 10 > gpurun_out/stress_final.log 2>&1; echo "stress rc=$?"; tail -1 gpurun_out/stress_final.log
timeout 300 python scripts/graph_pair_probe.py > gpurun_out/graph_pair_probe.txt 2>&1; echo "graph pair probe rc=$?"; tail -3 gpurun_out/graph_pair_probe.txt | cut -c1-250
( timeout 400 python scripts/decode_chain_probe.py --batch 1 --short; timeout 400 python scripts/decode_chain_probe.py --batch 64 --short ) > gpurun_out/decode_chain_probe_short.txt 2>&1; echo "chain probe rc=$?"; grep "us per token" gpurun_out/decode_chain_probe_short.txt
bash scripts/dp_world1_ab.sh asr > gpurun_out/dp_world1_ab_asr.txt 2>&1; echo "dp world-1 rc=$?"; tail -4 gpurun_out/dp_world1_ab_asr.txt | cut -c1-120
bash scripts/gpu_timeline.sh --workload asr --sustain-s 0.2 --no-decode > gpurun_out/tl_asr.txt 2>&1; python profiles/timeline.py gpurun_out/prof_graph/bench_results.db 3 --share > gpurun_out/asr_timeline_share.txt 2>&1; rm -rf gpurun_out/prof_graph gpurun_out/prof gpurun_out/prof_dec_b1 gpurun_out/prof_dec_b64; head -3 gpurun_out/asr_timeline_share.txt | cut -c1-200
du -sh gpurun_out
