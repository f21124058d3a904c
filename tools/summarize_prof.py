"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table for profiles/.
usage: python tools/summarize_prof.py gpurun_out/prof1 profiles/r01_bench_kernel_stats.csv [steps]"""
import csv, glob, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
path = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    mode = ("BMP_ONE_STREAM=1 (every launch whole and in line on one stream)" if os.environ.get("BMP_ONE_STREAM") == "1"
            else "default (weight-gradient launches on the low-priority side stream, forward as two chains of tiles: launches share the CUs)")
    f.write(f"# source: rocprofv3 --kernel-trace --stats -- python bench.py ... ; {mode}; kernel names shortened\n")
    f.write("name,calls,total_ms,avg_us,percent" + (",ms_per_step" if steps else "") + "\n")
    for r in rows[:40]:
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        name = name if len(name) < 70 else name[:67] + "..."
        line = f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.3f},{float(r["AverageNs"]) / 1e3:.2f},{float(r["Percentage"]):.2f}'
        if steps:
            line += f',{float(r["TotalDurationNs"]) / 1e6 / steps:.3f}'
        f.write(line + "\n")
    f.write(f"# all kernels: {tot / 1e6:.3f} ms" + (f" = {tot / 1e6 / steps:.3f} ms per step over {steps:g} steps\n" if steps else "\n"))
print(open(dst).read()[:1500])
