"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/.
usage: python profiles/summarize_rocpd.py gpurun_out/prof/bench_results.db [steps] > profiles/rNN_xxx.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary: {len(rows)} dispatches, {len(agg)} kernels, "
      f"kernel time {total / 1e6:.3f} ms, span {(t1 - t0) / 1e6:.3f} ms, per step (/{steps}): {total / 1e6 / steps:.3f} ms")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / total:6.2f}  {name}")
