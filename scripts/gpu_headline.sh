# headline records: default bench (AV configs[2]) + ASR configs[1] + PMC HBM passes + kernel table of the AV step
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -4 gpurun_out/bench_default.err
cut -c1-400 gpurun_out/bench_default.json
timeout 900 python bench.py --workload asr --steps 20 --warmup 5 > gpurun_out/bench_asr.json 2> gpurun_out/bench_asr.err; echo "asr rc=$?"
cut -c1-300 gpurun_out/bench_asr.json
OUT=pmc_hbm_av bash scripts/gpu_pmc_hbm.sh
bash scripts/gpu_prof.sh
