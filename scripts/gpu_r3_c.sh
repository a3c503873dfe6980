#!/bin/bash
# round 3: row-block streaming FFN kernel: parity, timing, PMC
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ffn2.py -x -q 2>&1 | tail -15 | tee gpurun_out/r3c_ffn2_tests.txt
timeout 600 python scripts/ffn2_bench.py 2>&1 | tee gpurun_out/r3c_ffn2_bench.txt
python scripts/ffn2_trace.py 10,4 10,3 8,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c_trace.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_a -o p --output-format csv -- python3 $R/scripts/ffn2_prof.py 10,4 10,3 > $R/gpurun_out/pmc_a.log 2>&1
echo "rc=$?"
cd $R
python profiles/summarize_pmc_mfma.py gpurun_out/pmc_a/p_counter_collection.csv gpurun_out/pmc_a/p_kernel_trace.csv > gpurun_out/r3c_pmc_mfma.txt; head -12 gpurun_out/r3c_pmc_mfma.txt | cut -c1-200
python - <<'PY'
import csv, re, collections
csv.field_size_limit(1 << 30)
dur = {}
for row in csv.DictReader(open("gpurun_out/pmc_a/p_kernel_trace.csv")):
    dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for row in csv.DictReader(open("gpurun_out/pmc_a/p_counter_collection.csv")):
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = re.sub(r"\(.*$", "", row["Kernel_Name"])[:60] + " grid=" + row.get("Grid_Size", "?") + " lds=" + row.get("LDS_Block_Size", "?")
    agg[k][0] += float(row["Counter_Value"]); agg[k][1] += dur.get(row["Dispatch_Id"], 0.0); agg[k][2] += 1
for k, (c, ns, n) in agg.items():
    if ns > 0 and n >= 6: print(f"clock {c / 8 / ns:5.2f} GHz  {ns / n / 1e3:7.1f} us x{n}  {k}")
PY
rm -rf gpurun_out/pmc_a
timeout 900 python bench.py --mode fwd-encoder > gpurun_out/r3c_fwd_encoder_ffn2.json 2>gpurun_out/r3c_fwd_encoder_ffn2.err
python - <<'PY'
import json
for n in ("ffn2",):
    try:
        d = json.loads(open(f"gpurun_out/r3c_fwd_encoder_{n}.json").read().strip().splitlines()[-1])["fwd_encoder"]
        print(n, {k: v for k, v in d.items() if k.startswith("layers12")})
    except Exception as e:
        print(n, "failed", e)
PY
