# usage: [OUT=name] bash scripts/gpu_pmc_mfma.sh [bench.py flags, e.g. --workload asr | --mode fwd-encoder]
# MFMA-utilisation counters of the benchmark step, per kernel (north_star: "evidenced by rocprof ... MFMA utilisation"):
# one rocprofv3 pass, --pmc with --kernel-trace only, the program directly after "--" (no env / bash -c hop).
#   SQ_BUSY_CU_CYCLES            CU-cycles with a wave resident (summed over the XCDs' SQs)
#   SQ_VALU_MFMA_BUSY_CYCLES     cycles the MFMA pipe is busy
#   SQ_INSTS_VALU_MFMA_MOPS_F32  fp32 MFMA operations issued (units of 512 FLOP)
#   SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY: where the waves' cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fwd-encoder --no-eager --no-asr --no-box --sustain-s 0 "$@" > $R/gpurun_out/pmc_mfma.log 2>&1
echo "rc=$?"; grep -v "^    @" $R/gpurun_out/pmc_mfma.log | tail -2 | cut -c1-300
cd $R
OUT=${OUT:-pmc_mfma}
python profiles/summarize_pmc_mfma.py gpurun_out/pmc_mfma/p_counter_collection.csv gpurun_out/pmc_mfma/p_kernel_trace.csv > gpurun_out/$OUT.txt; head -24 gpurun_out/$OUT.txt | cut -c1-200
rm -rf gpurun_out/pmc_mfma
