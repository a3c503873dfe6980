"""Occupancy of the device timeline from a rocprofv3 kernel trace (rocpd sqlite) of the hipGraph-replayed bench:
busy time (>=1 kernel running), idle gaps between kernels, time with >=2 kernels overlapping, per replayed step.
usage: python profiles/timeline.py gpurun_out/prof_graph/bench_results.db [last_n_steps] [--kernels] [--share]
--share: wall-clock share per kernel name - every instant of a step is split evenly between the kernels running at it, an idle gap goes to
the kernel that starts behind it ("(gap before) name"): the rows add up to the step's span, which a sum of kernel times does not."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [re.sub(r"\(anonymous namespace\)::", "", r[0]) for r in rows]
names = [re.sub(r"\(.*$", "", n) for n in names]
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
if "--sequence" in sys.argv:     # the last replayed step, launch by launch: start offset, duration, queue, how many kernels run beside it at its start
    qrows = db.execute("select start, end, queue_id, stream_id from kernels order by start").fetchall() if "--sequence" in sys.argv else []
    lo, hi = marks[-2], marks[-1]
    t0 = rows[lo][1]
    ends = []
    for i in range(lo, hi):
        ends = [e for e in ends if e > rows[i][1]]
        print(f"{(rows[i][1] - t0) / 1e3:10.2f} us  {(rows[i][2] - rows[i][1]) / 1e3:8.2f} us  q{qrows[i][2]} s{qrows[i][3]}  beside {len(ends)}  {names[i][:120]}")
        ends.append(rows[i][2])
    sys.exit(0)
if "--share" in sys.argv:
    share, gapto, calls = {}, {}, {}
    span = 0
    for si in range(max(0, len(marks) - 1 - nlast), len(marks) - 1):
        seg = list(zip(rows[marks[si]:marks[si + 1]], names[marks[si]:marks[si + 1]]))
        ev = sorted([(r[1], 1, i) for i, (r, _) in enumerate(seg)] + [(r[2], 0, i) for i, (r, _) in enumerate(seg)])
        live = set()
        last = ev[0][0]
        span += max(r[2] for r, _ in seg) - last
        for t, opening, i in ev:
            if live:
                for j in live:
                    share[seg[j][1]] = share.get(seg[j][1], 0) + (t - last) / len(live)
            elif opening and t > last:
                gapto[seg[i][1]] = gapto.get(seg[i][1], 0) + (t - last)
            if opening:
                live.add(i)
                calls[seg[i][1]] = calls.get(seg[i][1], 0) + 1
            else:
                live.discard(i)
            last = t
    print(f"# wall-clock share over the last {nlast} replayed steps: span {span / 1e6 / nlast:.3f} ms/step, in kernels {sum(share.values()) / 1e6 / nlast:.3f}, "
          f"in gaps {sum(gapto.values()) / 1e6 / nlast:.3f}")
    for nm in sorted(share, key=lambda k: -(share[k] + gapto.get(k, 0)))[:45]:
        print(f"{calls[nm] / nlast:8.1f} calls/step  share {share[nm] / 1e6 / nlast:7.3f} ms/step  + gap before {gapto.get(nm, 0) / 1e6 / nlast:6.3f} ms/step   {nm[:110]}")
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
