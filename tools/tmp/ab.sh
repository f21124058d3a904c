cd /root/repo
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for c in c2 c3; do
for v in 1 1; do
  timeout -k 10 300 python bench.py --config $c --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])" || exit 1
done; done
