set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/t_gpu.log 2>&1 || (tail -40 gpurun_out/t_gpu.log)
tail -3 gpurun_out/t_gpu.log
for p in mixed normal; do
  python tools/stream_cliff.py --prio $p > gpurun_out/cliff_$p.log 2>&1
  GPU_MAX_HW_QUEUES=8 python tools/stream_cliff.py --prio $p > gpurun_out/cliff_${p}_q8.log 2>&1
done
GPU_MAX_HW_QUEUES=2 python tools/stream_cliff.py --prio mixed > gpurun_out/cliff_mixed_q2.log 2>&1
tail -1 gpurun_out/cliff_*.log
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
$B > gpurun_out/b_c2_chains2.json 2> gpurun_out/b_c2_chains2.err
BMP_FWD_CHAINS=4 $B > gpurun_out/b_c2_chains4.json 2> gpurun_out/b_c2_chains4.err
BMP_FWD_CHAINS=4 GPU_MAX_HW_QUEUES=8 $B > gpurun_out/b_c2_chains4_q8.json 2> gpurun_out/b_c2_chains4_q8.err
BMP_FWD_CHAINS=3 $B > gpurun_out/b_c2_chains3.json 2> gpurun_out/b_c2_chains3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c2_chains*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/tr_chains4 -- python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 3 > gpurun_out/tr4.log 2>&1 || true
BMP_FWD_CHAINS=4 rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/tr_chains4b -- python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 3 > gpurun_out/tr4b.log 2>&1 || true
ls -la gpurun_out/tr_chains4* | head; du -sh gpurun_out/tr_chains4 gpurun_out/tr_chains4b
