set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd _r02 && python bench.py --no-cpu-baseline --steps 20 --warmup 4 > ../gpurun_out/bench_r02tree.json 2> ../gpurun_out/bench_r02tree.err)
BMP_BENCH_OTHERS=0 python bench.py --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
(cd _r02 && python bench.py --no-cpu-baseline --steps 20 --warmup 4 > ../gpurun_out/bench_r02tree2.json 2> ../gpurun_out/bench_r02tree2.err)
python - <<'PY'
import json
for f in ("bench_r02tree", "bench_now", "bench_r02tree2"):
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d["value"], {k: v for k, v in d["batch32"].items() if k != "what"}, d["end_to_end"]["collate_call_ms_per_batch"], d["end_to_end"]["value"])
PY
