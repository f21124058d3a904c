"""Condense rocprofv3 --pmc counter_collection.csv files into per-kernel means (one JSON for profiles/).
usage: python tools/summarize_pmc.py OUT.json DIR [DIR ...]      (every DIR = the -d of one --pmc pass)
Per kernel (template arguments kept, parameter list dropped): launches, mean duration (us, under the profiler)
and the mean value per launch of every counter collected in any of the passes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dd = dur[name]
                dd[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dd[1] += 1
res = {}
for name, cs in acc.items():
    if not name.startswith("k_"):
        continue
    e = {"launches": max(v[1] for v in cs.values()), "avg_us_profiled": round(dur[name][0] / max(dur[name][1], 1), 2)}
    for c, (s, n) in sorted(cs.items()):
        e[c + "_per_launch"] = s / n
    res[name] = e
# provenance for bench.py: traffic figures are only quoted for the library version and configuration they were measured on
import subprocess
tag = {"_config": os.environ.get("BMP_PROFILE_CONFIG", "c2")}
try:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gcn-bmp_amd"))
    from bmp import _lib
    tag["_bmp_version"] = int(_lib.lib().bmp_version())
except Exception as e:                               # noqa: BLE001
    tag["_bmp_version"] = None
res.update(tag)
json.dump(res, open(out, "w"), indent=1)
for name, e in sorted(((k, v) for k, v in res.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["avg_us_profiled"] * kv[1]["launches"])[:12]:
    print(name, json.dumps(e))
