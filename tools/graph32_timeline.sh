# Kernel timeline of one replay of the recorded 32-pair step (tools/graph32_probe.py under rocprofv3 --kernel-trace):
#   /usr/local/graft/bin/gpurun -- bash tools/graph32_timeline.sh   -> gpurun_out/timeline_g32.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_g32 -- python tools/graph32_probe.py > gpurun_out/kt_g32.log 2>&1
python tools/step_timeline.py gpurun_out/kt_g32 60 > gpurun_out/timeline_g32.txt
rm -rf gpurun_out/kt_g32
