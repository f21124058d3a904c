# Round-4 profiles: kernel stats (one stream + overlapped), PMC passes (HBM traffic for c2 / c3 / c4, SQ for c2 / c3), bench lines.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh'   -> gpurun_out/r04_* (copy into profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r04
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 4"
# kernel stats with every launch whole and in line on one stream (BMP_ONE_STREAM=1): a kernel's duration is its own.  In the
# default run the weight-gradient launches share the CUs with the backward chain (low-priority side stream), the forward
# runs as two chains of tiles, and traced durations span the sharing: that trace is kept for C2 as *_overlapped.csv
export BMP_ONE_STREAM=1
for c in c2 c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$c -- $B --config $c > gpurun_out/ks_$c.log 2>&1
  python tools/summarize_prof.py gpurun_out/ks_$c gpurun_out/${R}_${c}_kernel_stats.csv 34 > /dev/null
  rm -rf gpurun_out/ks_$c
done
# counter passes: whole launches too (per-launch counters next to the roofline leg's per-launch figures); FETCH and WRITE in
# separate passes, as MI355X_MICROARCH.md prescribes
for c in c2 c3 c4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- $B --config $c > gpurun_out/pmc_f_$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- $B --config $c > gpurun_out/pmc_w_$c.log 2>&1
  BMP_PROFILE_CONFIG=$c python tools/summarize_pmc.py gpurun_out/${R}_${c}_pmc_hbm_traffic.json gpurun_out/pmc_f gpurun_out/pmc_w > /dev/null
  rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
done
for c in c2 c3 c4; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/pmc_s -- $B --config $c > gpurun_out/pmc_s_$c.log 2>&1
  BMP_PROFILE_CONFIG=$c python tools/summarize_pmc.py gpurun_out/${R}_${c}_pmc_sq.json gpurun_out/pmc_s > /dev/null
  rm -rf gpurun_out/pmc_s
done
unset BMP_ONE_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_o -- $B --config c2 > gpurun_out/ks_o.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_o gpurun_out/${R}_c2_kernel_stats_overlapped.csv 34 > /dev/null
rm -rf gpurun_out/ks_o
ls gpurun_out | grep ${R}_
cp gpurun_out/${R}_c?_pmc_hbm_traffic.json profiles/      # the bench quotes `traffic` from the files of ITS library version
# the reference's published model (d = 32 x 8 untied + Nie + NTN): kernel stats of its step
export BMP_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_ref -- $B --config ref_ntn > gpurun_out/ks_ref.log 2>&1
python tools/summarize_prof.py gpurun_out/ks_ref gpurun_out/${R}_ref_ntn_kernel_stats.csv 34 > /dev/null
rm -rf gpurun_out/ks_ref
unset BMP_ONE_STREAM
python bench.py > gpurun_out/${R}_bench_c2.json 2> gpurun_out/${R}_bench_c2.err
for c in c3 c4; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/${R}_bench_$c.json 2> gpurun_out/${R}_bench_$c.err
done
ls gpurun_out | grep ${R}_bench
