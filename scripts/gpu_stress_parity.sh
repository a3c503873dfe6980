#!/bin/bash
# parity checks while another process keeps the GPU busy (timing perturbation: looks for latent ordering hazards in the kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 170 python bench.py --workload asr --steps 4000 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --sustain-s 0 > /dev/null 2>&1 ) &
BG=$!
sleep 25
for i in 1 2 3; do python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; done
timeout 300 python -m pytest tests/test_beam_search.py -q -m gpu -k "matches_oracle" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffn2.py tests/test_gpu_gemm.py -q -m gpu -x 2>&1 | tail -2
kill $BG 2>/dev/null; wait $BG 2>/dev/null
echo done
