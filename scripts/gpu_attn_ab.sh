mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_attn.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --workload asr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager 2>/dev/null | cut -c100-200
bash scripts/gpu_prof.sh --workload asr | tail -2; grep attn_ gpurun_out/prof_stats.txt | cut -c1-130
