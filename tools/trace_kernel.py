"""List the durations of one kernel's launches from a rocprofv3 --kernel-trace csv.  usage: trace_kernel.py DIR NAME_SUBSTR"""
import csv, glob, os, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [r for r in csv.DictReader(open(path)) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; last 24 (us):", " ".join(f"{x:.0f}" for x in d[-24:]))
