set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
$B > gpurun_out/b_c2_chains2.json 2> gpurun_out/b_c2_chains2.err
BMP_FWD_CHAINS=4 $B > gpurun_out/b_c2_chains4.json 2> gpurun_out/b_c2_chains4.err
BMP_FWD_CHAINS=4 GPU_MAX_HW_QUEUES=8 $B > gpurun_out/b_c2_chains4_q8.json 2> gpurun_out/b_c2_chains4_q8.err
BMP_FWD_CHAINS=4 GPU_MAX_HW_QUEUES=2 $B > gpurun_out/b_c2_chains4_q2.json 2> gpurun_out/b_c2_chains4_q2.err
BMP_FWD_CHAINS=3 $B > gpurun_out/b_c2_chains3.json 2> gpurun_out/b_c2_chains3.err
BMP_FWD_CHAINS=4 BMP_WGRAD_STREAM=0 $B > gpurun_out/b_c2_chains4_noside.json 2> gpurun_out/b_c2_chains4_noside.err
BMP_FWD_CHAINS=4 BMP_COLLATE_STREAM=0 $B > gpurun_out/b_c2_chains4_nocollate.json 2> gpurun_out/b_c2_chains4_nocollate.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c2_chains*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/tr_chains2 -- python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 3 > gpurun_out/tr2.log 2>&1 || true
BMP_FWD_CHAINS=4 rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/tr_chains4 -- python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 3 > gpurun_out/tr4.log 2>&1 || true
du -sh gpurun_out/tr_chains2 gpurun_out/tr_chains4 || true
find gpurun_out/tr_chains4 -name "*.csv" | head
