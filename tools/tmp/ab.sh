cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_collate.py tests/test_gpu_fullsize.py tests/test_gpu_planned_oracle.py -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for c in c2 c3; do
for v in 0 1 0 1; do
  BMP_COLLATE_STREAM=$v timeout -k 10 400 python bench.py --config $c --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print('$c cs=$v', d['value'], d['ms_per_step'], 'e2e', e['value'], e['ratio_to_resident'], e['collate_call_ms_per_batch'], 'b32', d.get('batch32') and d['batch32']['value'])" || exit 1
done; done
