mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_av.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_av_q.json 2> gpurun_out/bench_av_q.err; echo rc=$?
cut -c1-330 gpurun_out/bench_av_q.json
bash scripts/gpu_prof.sh 2>&1 | grep -i "im2col\|rc=\|summary"
