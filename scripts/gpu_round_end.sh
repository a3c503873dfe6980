# round-end records: default bench (AV) + ASR bench, forward-encoder bench, kernel tables, per-shape GEMM tables, PMC passes
# (HBM bytes, MFMA utilisation), decode bench.  usage: bash scripts/gpu_round_end.sh   (outputs under gpurun_out/)
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -4 gpurun_out/bench_default.err
cut -c1-330 gpurun_out/bench_default.json
timeout 900 python bench.py --workload asr --steps 20 --warmup 5 > gpurun_out/bench_asr.json 2> gpurun_out/bench_asr.err; echo "asr rc=$?"
cut -c1-300 gpurun_out/bench_asr.json
timeout 600 python profiles/gemm_shapes.py --workload avsr > gpurun_out/gemm_shapes_av.txt 2>&1; echo "shapes av rc=$?"
timeout 600 python profiles/gemm_shapes.py --workload asr > gpurun_out/gemm_shapes_asr.txt 2>&1; echo "shapes asr rc=$?"
bash scripts/gpu_prof.sh | tail -3; cp gpurun_out/prof_stats.txt gpurun_out/kernel_stats_av.txt
bash scripts/gpu_prof.sh --workload asr | tail -3; cp gpurun_out/prof_stats.txt gpurun_out/kernel_stats_asr.txt
OUT=pmc_hbm_av bash scripts/gpu_pmc_hbm.sh | tail -4
OUT=pmc_hbm_asr bash scripts/gpu_pmc_hbm.sh --workload asr | tail -4
OUT=pmc_mfma_av bash scripts/gpu_pmc_mfma.sh | tail -4
OUT=pmc_mfma_asr bash scripts/gpu_pmc_mfma.sh --workload asr | tail -4
OUT=pmc_mfma_fwd_encoder bash scripts/gpu_pmc_mfma.sh --mode fwd-encoder | tail -4
if [ -z "$SKIP_DECODE" ]; then
timeout 900 python bench_decode.py --utterances 256 --batch 64 > gpurun_out/decode_b64.json 2> gpurun_out/decode_b64.err; echo "decode rc=$?"; cut -c1-300 gpurun_out/decode_b64.json
timeout 900 python bench_decode.py --utterances 16 --batch 1 --no-cpu-baseline > gpurun_out/decode_b1.json 2> gpurun_out/decode_b1.err; echo "decode1 rc=$?"; cut -c1-300 gpurun_out/decode_b1.json
timeout 900 python bench_decode.py --utterances 512 --batch 256 --no-cpu-baseline > gpurun_out/decode_b256.json 2> gpurun_out/decode_b256.err; echo "decode256 rc=$?"; cut -c1-300 gpurun_out/decode_b256.json
bash scripts/gpu_decode_prof.sh 1 | head -12
bash scripts/gpu_decode_prof.sh 64 | head -12
fi
timeout 600 python scripts/fwd_breakdown.py > gpurun_out/fwd_breakdown.txt 2>&1; head -14 gpurun_out/fwd_breakdown.txt | cut -c1-150
timeout 300 python scripts/ffn2_bench.py > gpurun_out/ffn2_bench.txt 2>&1; tail -12 gpurun_out/ffn2_bench.txt
timeout 300 python scripts/ffn2_bwd_bench.py > gpurun_out/ffn2_bwd_bench.txt 2>&1; tail -8 gpurun_out/ffn2_bwd_bench.txt
timeout 300 python scripts/attn_bench.py > gpurun_out/attn_bench.txt 2>&1; tail -6 gpurun_out/attn_bench.txt
timeout 300 python scripts/merge_bench.py > gpurun_out/merge_bench.txt 2>&1; tail -3 gpurun_out/merge_bench.txt
timeout 300 python scripts/eager_host_profile.py > gpurun_out/eager_host_profile.txt 2>&1; head -5 gpurun_out/eager_host_profile.txt
du -sh gpurun_out
