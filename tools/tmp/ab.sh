cd /root/repo
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for c in c4; do
for v in 0 1 0 1; do
  BMP_FWD_SPLIT=$v timeout -k 10 300 python bench.py --config $c --no-extras --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c split=$v', d['value'], d['ms_per_step'])" || exit 1
done; done
