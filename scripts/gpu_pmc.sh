cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc1 -o p --output-format csv -- python3 $R/profiles/gemm_pmc.py > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o p --output-format csv -- python3 $R/profiles/gemm_pmc.py > $R/gpurun_out/pmc2.log 2>&1
ls -R $R/gpurun_out/pmc1 | head; tail -3 $R/gpurun_out/pmc1.log $R/gpurun_out/pmc2.log
