cd /root/repo
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
/usr/bin/time -v timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; grep -E "Elapsed|Maximum resident" gpurun_out/final_bench.err; cut -c1-400 gpurun_out/final_bench.json
