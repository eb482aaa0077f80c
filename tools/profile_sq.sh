# One --pmc pass of SQ counters over one step of the default bench.py workload (wave cycles, parked / issue-stalled / issuing cycles, LDS issue stalls
# and bank conflicts): where the fused kernels' time goes.  Run on the GPU box from the repo root.
set -e
REPO=$PWD; export TMPDIR=/tmp
rm -rf gpurun_out/prof_sq
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d $REPO/gpurun_out/prof_sq -o cube -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $REPO/gpurun_out/prof_sq.log 2>&1 )
DB=$(find gpurun_out/prof_sq -name "*.db" | head -1)
python3 tools/rocprof_summary.py counters $DB gpurun_out/r2_sq_counters.csv
head -12 gpurun_out/r2_sq_counters.csv
rm -rf gpurun_out/prof_sq
