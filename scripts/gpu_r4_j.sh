#!/bin/bash
# implicit 3x3 convolution: taps of a channel slab in consecutive K-steps.  Conv parity, A/B of the AV step against the previous
# build (gpurun_in/lib_old), HBM counter pass of the new build.
set -o pipefail
mkdir -p gpurun_out
OLD=$PWD/gpurun_in/lib_old/libtavsr_hip.so
python -m pytest tests/test_gpu_ops.py tests/test_gpu_av.py tests/test_gpu_blocks.py tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export TAVSR_LIB=$OLD; else unset TAVSR_LIB; fi
  python bench.py --no-cpu-baseline --no-box --no-fwd-encoder --no-asr --no-eager --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$lib av', d['value'], d['ms_per_step'], 'NT', r['by_kernel']['gemm_kernel<NT>'], 'TN', r['by_kernel']['gemm_kernel<TN>']['tflops'])"
done; done
unset TAVSR_LIB
OUT=pmc_hbm_av bash scripts/gpu_pmc_hbm.sh | tail -4
head -12 gpurun_out/pmc_hbm_av.txt | cut -c1-160
