set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k "all_steps or tile_ranges or fused" > gpurun_out/t_sel.log 2>&1 || (tail -60 gpurun_out/t_sel.log; exit 1)
tail -n 3 gpurun_out/t_sel.log
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/t_gpu.log 2>&1 || (tail -60 gpurun_out/t_gpu.log; exit 1)
tail -n 3 gpurun_out/t_gpu.log
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 6"
$B > gpurun_out/b_c2_ts.json 2> gpurun_out/b_c2_ts.err
BMP_TSTEPS=0 $B > gpurun_out/b_c2_nots.json 2> gpurun_out/b_c2_nots.err
$B > gpurun_out/b_c2_ts2.json 2> gpurun_out/b_c2_ts2.err
BMP_TSTEPS=0 $B > gpurun_out/b_c2_nots2.json 2> gpurun_out/b_c2_nots2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c2_*ts*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['whole_step']['f32_frac'])
PY
BMP_BENCH_OTHERS=0 python bench.py --no-cpu-baseline > gpurun_out/bench_ts.json 2> gpurun_out/bench_ts.err
BMP_TSTEPS=0 BMP_BENCH_OTHERS=0 python bench.py --no-cpu-baseline > gpurun_out/bench_nots.json 2> gpurun_out/bench_nots.err
python - <<'PY'
import json
for L in ("ts", "nots"):
    d = json.loads(open(f'gpurun_out/bench_{L}.json').read().strip().splitlines()[-1])
    print("==", L, d["value"], d["end_to_end"]["value"], d["batch32"]["value"], d["predict"]["value"], d["dedup"]["value"], d["dedup"]["predict_value"])
PY
