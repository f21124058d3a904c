"""Timeline of a rocprofv3 --kernel-trace CSV: per kernel name its average duration and the average idle gap of its queue before
it starts (start - end of the previous dispatch on the same queue), per queue the busy and idle time, and over all queues the
time in which NO kernel runs.  Steps are taken between successive `k_adam` launches (the last `n_steps` of them).
   python tools/chain_gaps.py <dir or kernel_trace.csv> [n_steps]"""
import csv, glob, os, re, sys
from collections import defaultdict

src = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), re.sub(r"\(.*", "", r["Kernel_Name"])))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[3].startswith("k_adam")]
adam = adam[::-1][::int(os.environ.get("GAPS_ADAMS_PER_STEP", "1"))][::-1]          # a step may hold several optimizer launches
if len(adam) < n_steps + 1:
    sys.exit(f"only {len(adam)} k_adam launches")
lo, hi = adam[-n_steps - 1] + 1, adam[-1] + 1
win = rows[lo:hi]
t0, t1 = rows[lo - 1][1], win[-1][1]
span = (t1 - t0) / n_steps
print(f"{n_steps} steps, {span / 1e3:.1f} us per step (k_adam end to k_adam end)")
# union of busy intervals over all queues
busy, cur_s, cur_e = 0, None, None
for s, e, q, n in sorted(win):
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running: {busy / n_steps / 1e3:.1f} us per step; no kernel on any queue: {(t1 - t0 - busy) / n_steps / 1e3:.1f} us per step")
last_end = {}
per = defaultdict(lambda: [0, 0, 0, 0])          # calls, duration, gap before (same queue), global gap before
qbusy = defaultdict(int)
prev_global_end = t0
for s, e, q, n in win:
    p = per[(q, n)]
    p[0] += 1; p[1] += e - s
    if q in last_end:
        p[2] += max(0, s - last_end[q])
    p[3] += max(0, s - prev_global_end)
    prev_global_end = max(prev_global_end, e)
    last_end[q] = e
    qbusy[q] += e - s
for q in sorted(qbusy):
    print(f"queue {q}: busy {qbusy[q] / n_steps / 1e3:.1f} us per step")
print(f"{'queue':5s} {'kernel':58s} {'calls/step':>10s} {'avg us':>8s} {'gap before (queue)':>18s} {'nothing running':>16s}   per step: dur / qgap / idle")
for (q, n), (c, d, g, gg) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{q:5d} {n[:58]:58s} {c / n_steps:10.2f} {d / c / 1e3:8.1f} {g / c / 1e3:18.1f} {gg / c / 1e3:16.1f}   {d / n_steps / 1e3:7.1f} {g / n_steps / 1e3:7.1f} {gg / n_steps / 1e3:7.1f}")
if os.environ.get("GAPS_TIMELINE", "1") != "0":
    a, b = adam[-2] + 1, adam[-1] + 1
    z = rows[a - 1][1]
    print("\nlast step, launches in start order (us after the previous k_adam's end): queue start end name")
    for s, e, q, n in rows[a:b]:
        print(f"  q{q} {(s - z) / 1e3:8.1f} {(e - z) / 1e3:8.1f}  {n[:70]}")
