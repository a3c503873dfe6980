# Phase timeline of the GEMM kernel from the instrumented build (built by scripts/build_trace.sh before gpurun).
mkdir -p gpurun_out
TAVSR_LIB=$GRAFT_REPO_ROOT/tailored-avsr_amd/tavsr/lib_trace/libtavsr_hip.so timeout 600 python profiles/gemm_trace.py > gpurun_out/gemm_trace.txt 2>&1; echo "trace rc=$?"
tail -5 gpurun_out/gemm_trace.txt
