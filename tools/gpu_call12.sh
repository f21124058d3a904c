set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/t_gpu.log 2>&1 || (tail -60 gpurun_out/t_gpu.log; exit 1)
tail -n 3 gpurun_out/t_gpu.log
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 6"
for c in c2 c3; do
  $B --config $c > gpurun_out/b_${c}_now.json 2> gpurun_out/b_${c}_now.err
done
BMP_BENCH_FORCE_PG=1 $B > gpurun_out/b_c2_pg1.json 2> gpurun_out/b_c2_pg1.err
BMP_BENCH_FORCE_PG=1 BMP_FWD_SPLIT=0 $B > gpurun_out/b_c2_pg1_nosplit.json 2> gpurun_out/b_c2_pg1_nosplit.err
BMP_BENCH_ONE_DEVICE=1 BMP_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-extras --steps 20 --warmup 4 > gpurun_out/b_c2_2ranks.json 2> gpurun_out/b_c2_2ranks.err || (tail -20 gpurun_out/b_c2_2ranks.err; true)
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c?_now.json') + glob.glob('gpurun_out/b_c2_pg1*.json') + ['gpurun_out/b_c2_2ranks.json']):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('rank_ms_per_step'), d.get('dist'))
    except Exception as e: print(f, 'ERR', e)
PY
