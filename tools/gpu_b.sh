set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_plan.py tests/test_gpu_fullsize.py -x -q > gpurun_out/b_tests.log 2>&1 || { tail -n 40 gpurun_out/b_tests.log; exit 1; }
tail -n 3 gpurun_out/b_tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10"
for c in c2 c3; do
  $B --config $c > gpurun_out/b_${c}_h3.json 2> gpurun_out/b_${c}_h3.err
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_c2 -- $B --config c2 > gpurun_out/kt_c2.log 2>&1
python tools/chain_gaps.py gpurun_out/kt_c2 20 > gpurun_out/gaps_c2.txt
rm -rf gpurun_out/kt_c2
