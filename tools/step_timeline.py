"""One training step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue and kernel of every
dispatch between two successive `k_adam` launches (the step before the last one).
   python tools/step_timeline.py <dir or kernel_trace.csv> [step_from_end]"""
import csv, glob, os, re, sys

src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), re.sub(r"\(.*", "", r["Kernel_Name"]),
                     r.get("Grid_Size_X", "") or r.get("Grid_Size", ""), r.get("Workgroup_Size_X", "") or r.get("Workgroup_Size", "")))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[3].startswith("k_adam")]
lo, hi = adam[-back - 1], adam[-back]
t0 = rows[lo][1]
qs = {}
print(f"step of {(rows[hi][1] - t0) / 1e3:.1f} us")
for s, e, q, n, g, w in rows[lo + 1:hi + 1]:
    qi = qs.setdefault(q, len(qs))
    wg = (int(g) // int(w)) if g and w and int(w) else 0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qi}  {'  ' * qi}{n[:70]}  [{wg}]")
