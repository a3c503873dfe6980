"""MFMA utilisation per kernel from a rocprofv3 `--kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv`
pass (scripts/gpu_pmc_mfma.sh).

Columns (per kernel, summed over its dispatches):
  mfma_busy/cu_busy   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES: share of the CU-resident cycles in which the MFMA
                      pipe is busy (gfx94x formula of rocprof's MfmaUtil: the counter is summed over 4 SIMDs per CU,
                      so the ratio is divided by 4)
  mfma_TF/s           SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 FLOP / the kernel's wall time (kernel trace), and as a fraction
                      of the 157.3 TFLOP/s fp32 MFMA peak
  wait / active       SQ_WAIT_ANY and SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (parked on s_waitcnt / barrier; issuing)
usage: python profiles/summarize_pmc_mfma.py <p_counter_collection.csv> [<p_kernel_trace.csv>]"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
PEAK = 157.3e12
agg, dur = {}, {}
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
    a = agg.setdefault(name, {"disp": set()})
    a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    a["disp"].add(row["Dispatch_Id"])
    if "Start_Timestamp" in row and row.get("End_Timestamp"):
        dur.setdefault(name, {})[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
if len(sys.argv) > 2:
    try:
        for row in csv.DictReader(open(sys.argv[2])):
            name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
            dur.setdefault(name, {})[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    except FileNotFoundError:
        pass
rows = []
for name, a in agg.items():
    ns = sum(dur.get(name, {}).values())
    flop = a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
    busy, cu = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("SQ_BUSY_CU_CYCLES", 0.0)
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    rows.append((ns, len(a["disp"]), busy / cu / 4.0 if cu else 0.0, flop / ns * 1e9 if ns else 0.0, flop,
                 a.get("SQ_WAIT_ANY", 0.0) / wc if wc else 0.0, a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else 0.0, name))
tot_ns = sum(r[0] for r in rows)
tot_flop = sum(r[4] for r in rows)
print("# MFMA utilisation per kernel (rocprofv3 --pmc, one pass; see profiles/summarize_pmc_mfma.py for the formulas)")
print(f"# all kernels: {tot_ns / 1e6:.3f} ms of kernel time, {tot_flop / 1e9:.1f} GFLOP issued as fp32 MFMA "
      f"-> {tot_flop / tot_ns / 1e3 if tot_ns else 0:.1f} TFLOP/s over the summed kernel time "
      f"({tot_flop / tot_ns * 1e9 / PEAK if tot_ns else 0:.3f} of the fp32 MFMA peak)")
print(f"{'calls':>6} {'ms':>9} {'mfma_busy/cu_busy':>18} {'mfma_TF/s':>10} {'of_peak':>8} {'GFLOP':>10} {'wait':>6} {'active':>7}  kernel")
for ns, n, util, fps, flop, wait, act, name in sorted(rows, reverse=True)[:40]:
    print(f"{n:6d} {ns / 1e6:9.3f} {util:18.3f} {fps / 1e12:10.2f} {fps / PEAK:8.3f} {flop / 1e9:10.2f} {wait:6.2f} {act:7.2f}  {name[:100]}")
