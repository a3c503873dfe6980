# round-end verification and records: smoke, full GPU suite, default bench (AV) + ASR bench, kernel table, PMC HBM passes
mkdir -p gpurun_out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -4 gpurun_out/bench_default.err
cut -c1-330 gpurun_out/bench_default.json
timeout 900 python bench.py --workload asr --steps 20 --warmup 5 > gpurun_out/bench_asr.json 2> gpurun_out/bench_asr.err; echo "asr rc=$?"
cut -c1-300 gpurun_out/bench_asr.json
bash scripts/gpu_prof.sh | tail -3
OUT=pmc_hbm_av bash scripts/gpu_pmc_hbm.sh | tail -8
