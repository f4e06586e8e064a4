#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun): rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes, as MI355X_MICROARCH.md prescribes) and one SQ pass, for the headline batch (64 distinct streams), the
# 12 800-capture scanner batch, the 1 024-stream batch (throughput regime) and stream mode.  Raw .db files land in gpurun_out/;
# profiles/rocpd_summary.py turns them into the small files kept under profiles/.  The program itself follows `--`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams"
CAL="$B --steps 20 --warmup 3"
BIG="$B --steps 10 --warmup 3 --streams 1024"
STR="$B --steps 10 --warmup 3 --mode stream"
SCAN="$B --workload scan --streams 12800 --frames 64 --distinct 32 --steps 6 --warmup 2"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
$CAL > $R/gpurun_out/r03_bench_n1_noprof.json 2> /dev/null      # (fills the stream cache; also the unprofiled line)
run() { local name=$1; shift; rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1 || echo "rocprofv3 $name failed ($?)"; }
run r03_stats         --kernel-trace --stats -d $R/gpurun_out/r03_stats -o r03 -- $CAL
run r03_fetch         --pmc FETCH_SIZE -d $R/gpurun_out/r03_fetch -o r03 -- $CAL
run r03_write         --pmc WRITE_SIZE -d $R/gpurun_out/r03_write -o r03 -- $CAL
run r03_sq            --pmc $SQ -d $R/gpurun_out/r03_sq -o r03 -- $CAL
run r03_big_stats     --kernel-trace --stats -d $R/gpurun_out/r03_big_stats -o r03 -- $BIG
run r03_big_sq        --pmc $SQ -d $R/gpurun_out/r03_big_sq -o r03 -- $BIG
run r03_str_stats     --kernel-trace --stats -d $R/gpurun_out/r03_str_stats -o r03 -- $STR
run r03_str_sq        --pmc $SQ -d $R/gpurun_out/r03_str_sq -o r03 -- $STR
run r03_scan_stats    --kernel-trace --stats -d $R/gpurun_out/r03_scan_stats -o r03 -- $SCAN
run r03_scan_fetch    --pmc FETCH_SIZE -d $R/gpurun_out/r03_scan_fetch -o r03 -- $SCAN
run r03_scan_write    --pmc WRITE_SIZE -d $R/gpurun_out/r03_scan_write -o r03 -- $SCAN
run r03_scan_sq       --pmc $SQ -d $R/gpurun_out/r03_scan_sq -o r03 -- $SCAN
cd $R
P="python3 profiles/rocpd_summary.py"
db() { find gpurun_out/$1 -name '*.db' | head -1; }
$P stats $(db r03_stats) profiles/r03_kernel_stats.csv 3
$P pmc $(db r03_fetch) $(db r03_write) profiles/r03_pmc_traffic.json 64 1020000
$P sq $(db r03_sq) profiles/r03_sq_counters.csv
$P stats $(db r03_big_stats) profiles/r03_streams1024_kernel_stats.csv 3
$P sq $(db r03_big_sq) profiles/r03_streams1024_sq_counters.csv
$P stats $(db r03_str_stats) profiles/r03_stream_mode_kernel_stats.csv 3
$P sq $(db r03_str_sq) profiles/r03_stream_mode_sq_counters.csv
$P stats $(db r03_scan_stats) profiles/r03_scan12800_kernel_stats.csv 2
$P pmc $(db r03_scan_fetch) $(db r03_scan_write) profiles/r03_scan12800_pmc_traffic.json 12800 640000
$P sq $(db r03_scan_sq) profiles/r03_scan12800_sq_counters.csv
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_* gpurun_out/profiles_r03/
ls -la profiles/ | tail -14
tail -3 gpurun_out/r03_stats.log
