set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_relgcn.py tests/test_gpu_planned_oracle.py tests/test_gpu_fullsize_backward.py tests/test_gpu_fullsize.py tests/test_gpu_enclayout.py -q -x > gpurun_out/t_sel.log 2>&1 || (tail -60 gpurun_out/t_sel.log; exit 1)
tail -n 3 gpurun_out/t_sel.log
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 6"
for c in c2 c3; do
  $B --config $c > gpurun_out/b_${c}_wide.json 2> gpurun_out/b_${c}_wide.err
  BMP_WGRAD_WIDE=0 $B --config $c > gpurun_out/b_${c}_narrow.json 2> gpurun_out/b_${c}_narrow.err
  $B --config $c > gpurun_out/b_${c}_wide2.json 2> gpurun_out/b_${c}_wide2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c?_wide*.json') + glob.glob('gpurun_out/b_c?_narrow.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['whole_step']['f32_frac'])
    except Exception as e: print(f, 'ERR', e)
PY
export BMP_ONE_STREAM=1
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_w -- $B > gpurun_out/ks_w.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_w gpurun_out/ks_wide_stats.csv 34 > /dev/null
rm -rf gpurun_out/ks_w
