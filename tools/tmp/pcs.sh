cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d gpurun_out/pcs -- python bench.py --config c4 --no-extras --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/pcs.log 2>&1
echo rc=$?
tail -5 gpurun_out/pcs.log
find gpurun_out/pcs -type f | head; du -sh gpurun_out/pcs
