# Instrumented (-DTAVSR_GEMM_TRACE) build of the library for profiles/gemm_trace.py; the product build is untouched.
set -e
cd "$(dirname "$0")/../tailored-avsr_amd/csrc"
make -j8 > /dev/null
mkdir -p build_trace ../tavsr/lib_trace
hipcc -DTAVSR_GEMM_TRACE -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -c gemm.hip -o build_trace/gemm.o
objs=$(ls build/*.o | grep -v build/gemm.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tavsr/lib_trace/libtavsr_hip.so build_trace/gemm.o $objs
echo built ../tavsr/lib_trace/libtavsr_hip.so
