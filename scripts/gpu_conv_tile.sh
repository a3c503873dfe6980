# A/B of the convolution tile shapes on the headline step (TAVSR_CONV_TILE / TAVSR_CONV_DW_TILE: 0 = 64x64 everywhere)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/ct.json 2> gpurun_out/ct.err; echo "$* rc=$?"; python -c "import json;d=json.load(open('gpurun_out/ct.json'));print('  ',d['value'],d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_gpu_av.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
run TAVSR_CONV_TILE=0 TAVSR_CONV_DW_TILE=0
run TAVSR_CONV_TILE=1 TAVSR_CONV_DW_TILE=1
run TAVSR_CONV_TILE=0 TAVSR_CONV_DW_TILE=0
run TAVSR_CONV_TILE=1 TAVSR_CONV_DW_TILE=1
