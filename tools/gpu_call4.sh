set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || (tail -30 gpurun_out/bench_full.err; exit 1)
echo "bench wall seconds: $(( $(date +%s) - T0 ))"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
for k in ("value","ms_per_step","whole_step","end_to_end","batch32","predict","dedup"):
    v = d.get(k)
    if isinstance(v, dict): v = {a: b for a, b in v.items() if a != "what"}
    print(k, v)
print("roofline", {k: v for k, v in d["roofline"].items() if k in ("kernel","frac","achieved","avg_launch_us","traffic")})
for c, v in (d.get("other_configs") or {}).items():
    print(c, v["value"], v["ms_per_step"], v["whole_step"], v["dominant_kernel"])
PY
