"""Occupancy of the device timeline from a rocprofv3 kernel trace (rocpd sqlite) of the hipGraph-replayed bench:
busy time (>=1 kernel running), idle gaps between kernels, time with >=2 kernels overlapping, per replayed step.
usage: python profiles/timeline.py gpurun_out/prof_graph/bench_results.db [last_n_steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [re.sub(r"\(.*$", "", r[0]) for r in rows]
# steps are delimited by the rng_advance kernel that opens every step
marks = [i for i, n in enumerate(names) if "rng_advance" in n]
marks.append(len(rows))
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
