set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in encoder instance; do echo $L;
BMP_BENCH_LAYOUT=$L BMP_BENCH_OTHERS=0 python bench.py --no-cpu-baseline > gpurun_out/bench_$L.json 2> gpurun_out/bench_$L.err || (tail -30 gpurun_out/bench_$L.err; exit 1)
done
python - <<'PY'
import json
for L in ("encoder", "instance"):
    d = json.loads(open(f'gpurun_out/bench_{L}.json').read().strip().splitlines()[-1])
    print("==", L)
    for k in ("value","ms_per_step","end_to_end","batch32","predict","dedup"):
        v = d.get(k)
        if isinstance(v, dict): v = {a: b for a, b in v.items() if a != "what"}
        print(k, v)
PY
