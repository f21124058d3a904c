set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
export BMP_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_enc -- $B > gpurun_out/ks_enc.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_enc gpurun_out/ks_enc_stats.csv 34 > /dev/null
BMP_BENCH_LAYOUT=instance rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_inst -- $B > gpurun_out/ks_inst.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_inst gpurun_out/ks_inst_stats.csv 34 > /dev/null
rm -rf gpurun_out/ks_enc gpurun_out/ks_inst
