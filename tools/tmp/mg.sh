cd /root/repo
BMP_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 40 2>gpurun_out/mg1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force_pg rccl 1 rank', d['value'], d['ms_per_step'], d['n_gpus'])" || { tail -20 gpurun_out/mg1.err; exit 1; }
BMP_BENCH_ONE_DEVICE=1 BMP_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 4 --no-extras --no-cpu-baseline 2>gpurun_out/mg2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks one device gloo', d['value'], d['ms_per_step'], d['n_gpus'], d.get('rank_ms_per_step'))" || { tail -20 gpurun_out/mg2.err; exit 1; }
