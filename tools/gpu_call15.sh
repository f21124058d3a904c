set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/t_gpu.log 2>&1 || (tail -60 gpurun_out/t_gpu.log; exit 1)
tail -n 3 gpurun_out/t_gpu.log
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 6"
for c in c2 c3; do
  $B --config $c > gpurun_out/b_${c}_now.json 2> gpurun_out/b_${c}_now.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c?_now.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['whole_step']['f32_frac'])
PY
