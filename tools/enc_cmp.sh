# instance vs encoder layout of the 1024-pair step: kernel stats in line, throughput, and one step's timeline of each
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4 --config ${CFG:-c2}"
for lay in instance encoder; do
  BMP_BENCH_LAYOUT=$lay rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$lay -- $B > gpurun_out/kt_$lay.log 2>&1
  python tools/step_timeline.py gpurun_out/kt_$lay 3 > gpurun_out/timeline_$lay.txt
  rm -rf gpurun_out/kt_$lay
done
