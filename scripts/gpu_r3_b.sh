#!/bin/bash
# round 3: where does the streaming FFN kernel's time go?  same-tile timing probe + two PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== all units stream hidden tile 0 (weights L2-hot; results wrong, timing only)"
TAVSR_FFN2_DBG=1 python scripts/ffn2_bench.py 2>&1 | grep "M=3168 eval" | tee gpurun_out/r3b_sametile.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_a -o p --output-format csv -- python3 $R/scripts/ffn2_prof.py 256,3 256,4 > $R/gpurun_out/pmc_a.log 2>&1
echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_b -o p --output-format csv -- python3 $R/scripts/ffn2_prof.py 256,3 256,4 > $R/gpurun_out/pmc_b.log 2>&1
echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE -d $R/gpurun_out/pmc_c -o p --output-format csv -- python3 $R/scripts/ffn2_prof.py 256,3 256,4 > $R/gpurun_out/pmc_c.log 2>&1
echo "rc=$?"
cd $R
python profiles/summarize_pmc_mfma.py gpurun_out/pmc_a/p_counter_collection.csv gpurun_out/pmc_a/p_kernel_trace.csv > gpurun_out/r3b_pmc_mfma.txt; cat gpurun_out/r3b_pmc_mfma.txt | cut -c1-220
python - <<'PY'
import csv, re, collections
csv.field_size_limit(1 << 30)
for tag in ("pmc_b", "pmc_c"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    try:
        for row in csv.DictReader(open(f"gpurun_out/{tag}/p_counter_collection.csv")):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"])[:90] + " grid=" + row.get("Grid_Size", "?")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    except Exception as e:
        print(tag, "failed", e); continue
    for k, v in agg.items():
        print(tag, len(n[k]), k, {c: round(x / len(n[k]), 1) for c, x in v.items()})
PY
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
