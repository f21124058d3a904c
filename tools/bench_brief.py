"""Short view of a bench.py JSON line: python tools/bench_brief.py gpurun_out/x.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = lambda o, *ks: (g(o.get(ks[0]), *ks[1:]) if len(ks) > 1 else o.get(ks[0])) if isinstance(o, dict) else None
print("value", d["value"], "ms", d["ms_per_step"], "f32_frac", g(d, "whole_step", "f32_frac"))
r = d.get("roofline")
if r:
    print("roofline", r["kernel"][:40], r["avg_launch_us"], r["frac"])
    print({k[:34]: v for k, v in r["per_kernel_ms_per_step"].items()})
for k in ("end_to_end", "batch32", "predict", "dedup"):
    if d.get(k):
        print(k, {kk: vv for kk, vv in d[k].items() if kk in ("value", "ms_per_step", "ratio_to_resident", "host_ms_per_step", "speedup", "layout")})
for n, o in (d.get("other_configs") or {}).items():
    print(n, o["value"], o["ms_per_step"], o["whole_step"]["f32_frac"], g(o, "dominant_kernel", "kernel"), g(o, "dominant_kernel", "frac"))
    print("  ", {k[:34]: v for k, v in (o.get("per_kernel_ms_per_step") or {}).items()})
for n, o in (d.get("ref_headline") or {}).items():
    print(n, {k: {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "host_ms_per_step", "f32_frac", "fused_speedup", "unfused_value")} for k, v in o.items() if isinstance(v, dict)})
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
