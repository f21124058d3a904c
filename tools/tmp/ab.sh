cd /root/repo
export BMP_ONE_STREAM=1
BMP_GLDS_PROBE=1 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rowgemm_forms.py tests/test_gpu_planned_oracle.py tests/test_gpu_mlp.py -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for c in c4; do
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export BMP_GLDS_PROBE=1; else unset BMP_GLDS_PROBE; fi
  timeout -k 10 300 python bench.py --config $c --no-extras --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c one-stream glds=$v', d['value'], d['ms_per_step'])" || exit 1
done; done
