"""HBM traffic per kernel from a rocprofv3 `--pmc FETCH_SIZE WRITE_SIZE --output-format csv` pass
(scripts/gpu_pmc_hbm.sh).  Corrections per MI355X_MICROARCH.md "HBM": both counters are in KiB-units of 1024 B
as rocprofv3 reports them; on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) reads at 64 B, so it
is doubled; WRITE_SIZE is taken as is.
usage: python profiles/summarize_pmc.py <p_counter_collection.csv> [steps] [json_out] > profiles/rNN_pmc_hbm.txt"""
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = {}
for row in csv.DictReader(open(path)):
    name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
    a = agg.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "disp": set()})
    if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
        a[row["Counter_Name"]] += float(row["Counter_Value"])
        a["disp"].add(row["Dispatch_Id"])
out = {}
print(f"# HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE; FETCH_SIZE x2 for gfx950; KiB -> bytes), {steps} steps")
print(f"{'calls':>7} {'read_MB/launch':>15} {'write_MB/launch':>16} {'total_MB/step':>14}  kernel")
rows = []
for name, a in agg.items():
    n = max(1, len(a["disp"]))
    rd = 2.0 * a["FETCH_SIZE"] * 1024.0
    wr = a["WRITE_SIZE"] * 1024.0
    rows.append(((rd + wr) / steps, n, rd / n, wr / n, name))
    out[name] = {"calls": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n}
for tot, n, rd, wr, name in sorted(rows, reverse=True)[:40]:
    print(f"{n:7d} {rd / 1e6:15.3f} {wr / 1e6:16.3f} {tot / 1e6:14.2f}  {name[:110]}")
print(f"# all kernels: {sum(r[0] for r in rows) / 1e9:.3f} GB of HBM traffic per step")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=0, sort_keys=True)
