set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BMP_BENCH_OTHERS=0 python bench.py --no-cpu-baseline --host-profile --steps 20 --warmup 4 > gpurun_out/bench_hp.json 2> gpurun_out/bench_hp.err || (tail -30 gpurun_out/bench_hp.err; exit 1)
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_hp.json').read().strip().splitlines()[-1])
print(d["batch32"])
PY
