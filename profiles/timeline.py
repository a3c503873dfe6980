"""Occupancy of the device timeline from a rocprofv3 kernel trace (rocpd sqlite) of the hipGraph-replayed bench:
busy time (>=1 kernel running), idle gaps between kernels, time with >=2 kernels overlapping, per replayed step.
usage: python profiles/timeline.py gpurun_out/prof_graph/bench_results.db [last_n_steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [re.sub(r"\(.*$", "", r[0]) for r in rows]
# steps are delimited by the rng_advance kernel that opens every step
marks = [i for i, n in enumerate(names) if "rng_advance" in n]
marks.append(len(rows))
if "--kernels" in sys.argv:      # per-kernel totals over the last nlast replayed steps
    lo = marks[max(0, len(marks) - 1 - nlast)]
    agg = {}
    for (name, st, en), nm in zip(rows[lo:], names[lo:]):
        a = agg.setdefault(nm, [0, 0])
        a[0] += 1
        a[1] += en - st
    tot = sum(a[1] for a in agg.values())
    print(f"# last {nlast} replayed steps: kernel-time sum {tot / 1e6 / nlast:.3f} ms/step")
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
        print(f"{a[0] / nlast:8.1f} calls/step {a[1] / 1e6 / nlast:8.3f} ms/step {a[1] / a[0] / 1e3:8.2f} us  {nm[:100]}")
for si in range(max(0, len(marks) - 1 - nlast), len(marks) - 1):
    seg = rows[marks[si]:marks[si + 1]]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
    busy = over = 0
    depth = 0
    last = t0
    gaps = []
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        elif t > last:
            gaps.append(t - last)
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    ksum = sum(r[2] - r[1] for r in seg)
    gaps.sort()
    print(f"step {si}: {len(seg)} kernels, span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {sum(gaps) / 1e6:.3f} ms "
          f"({len(gaps)} gaps, median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us), >=2 kernels {over / 1e6:.3f} ms, kernel-time sum {ksum / 1e6:.3f} ms")
