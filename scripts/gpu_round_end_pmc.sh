# the profiler passes of scripts/gpu_round_end.sh alone (kernel tables, HBM and MFMA counters)
mkdir -p gpurun_out
bash scripts/gpu_prof.sh | tail -3; cp gpurun_out/prof_stats.txt gpurun_out/kernel_stats_av.txt
bash scripts/gpu_prof.sh --workload asr | tail -3; cp gpurun_out/prof_stats.txt gpurun_out/kernel_stats_asr.txt
OUT=pmc_hbm_av bash scripts/gpu_pmc_hbm.sh | tail -4
OUT=pmc_hbm_asr bash scripts/gpu_pmc_hbm.sh --workload asr | tail -4
OUT=pmc_mfma_av bash scripts/gpu_pmc_mfma.sh | tail -4
OUT=pmc_mfma_asr bash scripts/gpu_pmc_mfma.sh --workload asr | tail -4
OUT=pmc_mfma_fwd_encoder bash scripts/gpu_pmc_mfma.sh --mode fwd-encoder | tail -4
du -sh gpurun_out
