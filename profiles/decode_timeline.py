"""Per-token timeline of the beam search from a rocprofv3 kernel trace (rocpd sqlite) of bench_decode.py: a step is the
span between two beam_combine launches.  Prints the median step span, the time with at least one kernel running, and
per-kernel calls / time per step.   usage: python profiles/decode_timeline.py <results.db> [first_step last_step] [--sequence]
--sequence: one step in the middle, launch by launch (start offset, duration, hardware queue, kernels already running at its start)."""
import re
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [re.sub(r"\(.*$", "", r[0]).replace("void ", "").replace("tavsr::", "") for r in rows]
marks = [i for i, n in enumerate(names) if "beam_combine" in n or "beam_select" in n]
if "--sequence" in sys.argv:
    q = db.execute("select queue_id from kernels order by start").fetchall()
    si = len(marks) // 2
    t0, ends = rows[marks[si]][1], []
    for i in range(marks[si], marks[si + 1] + 1):
        ends = [e for e in ends if e > rows[i][1]]
        print(f"{(rows[i][1] - t0) / 1e3:9.2f} us  {(rows[i][2] - rows[i][1]) / 1e3:7.2f} us  q{q[i][0]}  beside {len(ends)}  {names[i][:100]}")
        ends.append(rows[i][2])
    sys.exit(0)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[2].isdigit() else (len(marks) // 2, len(marks) // 2 + 40)
hi = min(hi, len(marks) - 1)
spans, busys, overs, agg = [], [], [], {}
for si in range(lo, hi):
    seg = rows[marks[si]:marks[si + 1]]
    t0, t1 = seg[0][1], seg[-1][1]
    spans.append((rows[marks[si + 1]][1] - t0) / 1e3)
    ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
    depth, last, busy, over = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    busys.append(busy / 1e3)
    overs.append(over / 1e3)
    for (nm_, st, en), nm in zip(seg, names[marks[si]:marks[si + 1]]):
        a = agg.setdefault(nm, [0, 0])
        a[0] += 1
        a[1] += en - st
n = hi - lo
print(f"# steps {lo}..{hi}: median span {statistics.median(spans):.1f} us, busy {statistics.median(busys):.1f} us (>= 2 kernels at once: {statistics.median(overs):.1f} us), "
      f"{sum(a[0] for a in agg.values()) / n:.0f} launches/step, kernel-time sum {sum(a[1] for a in agg.values()) / n / 1e3:.1f} us/step")
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{a[0] / n:7.1f} calls/step {a[1] / n / 1e3:8.1f} us/step {a[1] / a[0] / 1e3:7.2f} us  {nm[:110]}")
