mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/ct.json 2> gpurun_out/ct.err; echo "$* rc=$?"; python -c "import json;d=json.load(open('gpurun_out/ct.json'));print('  ',d['value'],d['ms_per_step'])"; }
run TAVSR_STEM_DW_PLAN=0
run TAVSR_STEM_DW_PLAN=3,384
run TAVSR_STEM_DW_PLAN=3,512
run TAVSR_STEM_DW_PLAN=3,768
run TAVSR_STEM_DW_PLAN=0
run TAVSR_STEM_DW_PLAN=3,512
