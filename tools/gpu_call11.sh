set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py -q -x > gpurun_out/t_sel.log 2>&1 || (tail -60 gpurun_out/t_sel.log; exit 1)
tail -n 3 gpurun_out/t_sel.log
export BMP_ONE_STREAM=1
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
BMP_BENCH_LAYOUT=instance rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_inst -- $B > gpurun_out/ks_inst.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_inst gpurun_out/ks_inst_stats.csv 34 > /dev/null
rm -rf gpurun_out/ks_inst
unset BMP_ONE_STREAM
bash tools/gpu_call10.sh
