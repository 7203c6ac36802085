"""rocprofv3 --kernel-trace CSV of an eager bench run (side stream on) -> where one steady-state step's wall time goes: per HIP stream
(Queue_Id) busy time, the idle gaps between consecutive kernels on the main stream by size class, the time both streams run at once, and
the kernels of the step in launch order (optional dump).  usage: python tools/timeline.py <kernel_trace.csv> [dump.txt]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |bdgemm::", "", n)
    return re.sub(r"\(.*", "", n)[:70]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id") or r.get("Stream_Id") or "?", short(r["Kernel_Name"])))
rows.sort()
# steps: split at the optimizer kernel (one sgd_apply per step)
ends = [i for i, r in enumerate(rows) if r[3].startswith("sgd_apply")]
if len(ends) < 3:
    raise SystemExit("need >= 3 steps in the trace")
a, b = ends[-3] + 1, ends[-2] + 1            # the last-but-one full step
step = rows[a:b]
t0, t1 = step[0][0], step[-1][1]
print(f"step: {len(step)} launches, wall {(t1 - t0) / 1e6:.3f} ms")
byq = defaultdict(list)
for r in step:
    byq[r[2]].append(r)
main_q = max(byq, key=lambda q: len(byq[q]))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in rs)
    print(f"  queue {q}{' (main)' if q == main_q else ''}: {len(rs)} launches, busy {busy / 1e6:.3f} ms")
m = byq[main_q]
gaps = [(m[i + 1][0] - m[i][1], m[i][3], m[i + 1][3]) for i in range(len(m) - 1)]
cls = [(0, 2000), (2000, 5000), (5000, 20000), (20000, 10 ** 12)]
for lo, hi in cls:
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f"  main-stream gaps {lo / 1e3:g}-{hi / 1e3 if hi < 10 ** 11 else float('inf'):g} us: {len(g)} gaps, {sum(x[0] for x in g) / 1e6:.3f} ms")
neg = [x for x in gaps if x[0] < 0]
print(f"  (overlapping successors on the main stream: {len(neg)})")
# overlap of other queues with main
ev = []
for q, rs in byq.items():
    for s, e, _, _ in rs:
        ev.append((s, 1, q == main_q)); ev.append((e, -1, q == main_q))
ev.sort()
nm = ns = 0; last = t0; both = only_m = only_s = idle = 0
for t, d, ism in ev:
    dt = t - last
    if nm and ns: both += dt
    elif nm: only_m += dt
    elif ns: only_s += dt
    else: idle += dt
    last = t
    if ism: nm += d
    else: ns += d
print(f"  wall split: main only {only_m / 1e6:.3f}  side only {only_s / 1e6:.3f}  both {both / 1e6:.3f}  nothing running {idle / 1e6:.3f} ms")
small = [r for r in m if r[1] - r[0] < 10000]
print(f"  main-stream kernels shorter than 10 us: {len(small)}, {sum(e - s for s, e, _, _ in small) / 1e6:.3f} ms")
agg = defaultdict(lambda: [0, 0])
for s, e, _, n in small:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"      {c:4d} x {t / c / 1e3:6.1f} us  {n}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for s, e, q, n in step:
            f.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {'M' if q == main_q else 'S'} {n}\n")
