set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python profiles/gemm_shapes.py > gpurun_out/gemm_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
