set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r03
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
export BMP_ONE_STREAM=1
c=c3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$c -- $B --config $c > gpurun_out/ks_$c.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_$c gpurun_out/${R}_${c}_kernel_stats.csv 34 > /dev/null
rm -rf gpurun_out/ks_$c
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- $B --config $c > gpurun_out/pmc_f_$c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- $B --config $c > gpurun_out/pmc_w_$c.log 2>&1
BMP_PROFILE_CONFIG=$c python tools/summarize_pmc.py gpurun_out/${R}_${c}_pmc_hbm_traffic.json gpurun_out/pmc_f gpurun_out/pmc_w > /dev/null
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/pmc_s -- $B --config $c > gpurun_out/pmc_s_$c.log 2>&1
BMP_PROFILE_CONFIG=$c python tools/summarize_pmc.py gpurun_out/${R}_${c}_pmc_sq.json gpurun_out/pmc_s > /dev/null
rm -rf gpurun_out/pmc_s
unset BMP_ONE_STREAM
cp gpurun_out/${R}_c3_pmc_hbm_traffic.json profiles/
python bench.py > gpurun_out/${R}_bench_c2.json 2> gpurun_out/${R}_bench_c2.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/${R}_bench_c3.json 2> gpurun_out/${R}_bench_c3.err
python - <<'PY'
import json
for c in ("c2", "c3"):
    d = json.loads(open(f'gpurun_out/r03_bench_{c}.json').read().strip().splitlines()[-1])
    print(c, d["value"], d["ms_per_step"], d["whole_step"]["f32_frac"], d["roofline"]["frac"], d["roofline"]["traffic"], d["end_to_end"]["value"], d["end_to_end"]["ratio_to_resident"], (d.get("batch32") or {}).get("value"), d["predict"]["value"], (d.get("dedup") or {}).get("value"))
    for cc, v in (d.get("other_configs") or {}).items():
        print("  other", cc, v["value"], v["ms_per_step"], v["whole_step"]["f32_frac"], v["dominant_kernel"])
PY
