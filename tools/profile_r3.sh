# rocprofv3 passes of the default bench.py workload (C192L127 six faces, hydrostatic TL+AD) for profiles/: kernel stats, FETCH_SIZE and WRITE_SIZE
# in separate --pmc passes (MI355X_MICROARCH.md: the two do not fit one pass), then one pass of SQ counters.  Run on the GPU box from the repo root.
set -e
REPO=$PWD; export TMPDIR=/tmp
run() { ( cd /tmp && rocprofv3 "$@" ) ; }
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
run --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o cube -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
echo stats done
run --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_fetch -o cube -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
echo fetch done
run --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_write -o cube -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
echo write done
run --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d $REPO/gpurun_out/prof_sq -o cube -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_sq.log 2>&1
echo sq done
DBS=$(find gpurun_out/prof_stats -name "*.db" | head -1); DBF=$(find gpurun_out/prof_fetch -name "*.db" | head -1); DBW=$(find gpurun_out/prof_write -name "*.db" | head -1); DBQ=$(find gpurun_out/prof_sq -name "*.db" | head -1)
python3 tools/rocprof_summary.py stats $DBS gpurun_out/r3_kernel_stats.csv
python3 tools/rocprof_summary.py pmc $DBF $DBW gpurun_out/r3_pmc_traffic.csv
python3 tools/rocprof_summary.py counters $DBQ gpurun_out/r3_sq_counters.csv
head -8 gpurun_out/r3_kernel_stats.csv; head -8 gpurun_out/r3_pmc_traffic.csv; head -6 gpurun_out/r3_sq_counters.csv
# the databases are large: keep only the summaries
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
