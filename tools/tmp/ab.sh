cd /root/repo
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
bash tools/profile_round.sh
