mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "rc=$?"
TAVSR_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "rc=$?"
python profiles/gemm_shapes.py > gpurun_out/gemm_shapes.txt 2>&1
