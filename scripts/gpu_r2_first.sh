mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropout.py tests/test_train_harness.py tests/test_gpu_gemm.py -m gpu -x -q > gpurun_out/pytest_first.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_first.log
timeout 600 python bench.py --mode fwd-encoder --steps 20 --warmup 5 > gpurun_out/fwd_encoder.json 2> gpurun_out/fwd_encoder.err; echo "fwd rc=$?"; tail -3 gpurun_out/fwd_encoder.err; cat gpurun_out/fwd_encoder.json
timeout 600 python profiles/gemm_shapes.py --workload asr --fwd-only > gpurun_out/gemm_shapes_asr_fwd.txt 2>&1; echo "shapes rc=$?"; head -40 gpurun_out/gemm_shapes_asr_fwd.txt
OUT=pmc_mfma_fwd bash scripts/gpu_pmc_mfma.sh --mode fwd-encoder
