#!/bin/bash
# Round-4 profiles (run on the GPU box through gpurun), ONE refresh for the round's kept code: rocprofv3 kernel stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) and one SQ pass, for the headline batch (64
# distinct streams), the 12 800-capture scanner batch, the 1 024-stream batch (throughput regime) and stream mode; plus the
# unprofiled default bench line, the N > 1 step cost on one rank (tools/dist_cost.py) and the clock / LDS micro-benchmark.
# Raw .db files land in gpurun_out/; profiles/rocpd_summary.py turns them into the small files kept under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams"
CAL="$B --steps 20 --warmup 3"
CALC="$CAL --prewarm-steps 0"     # counter passes: the counters do not depend on the clock state, and 256 fewer steps keep the .db files small
BIG="$B --steps 10 --warmup 3 --streams 1024 --prewarm-steps 20"
STR="$B --steps 10 --warmup 3 --mode stream --prewarm-steps 20"
SCAN="$B --workload scan --streams 12800 --frames 64 --distinct 32 --steps 6 --warmup 2"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
$CAL > $R/gpurun_out/r04_bench_n1_noprof.json 2> /dev/null      # (fills the stream cache; also the unprofiled line)
run() { local name=$1; shift; rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1 || echo "rocprofv3 $name failed ($?)"; }
run r04_stats         --kernel-trace --stats -d $R/gpurun_out/r04_stats -o r04 -- $CAL
run r04_fetch         --pmc FETCH_SIZE -d $R/gpurun_out/r04_fetch -o r04 -- $CALC
run r04_write         --pmc WRITE_SIZE -d $R/gpurun_out/r04_write -o r04 -- $CALC
run r04_sq            --pmc $SQ -d $R/gpurun_out/r04_sq -o r04 -- $CALC
run r04_big_stats     --kernel-trace --stats -d $R/gpurun_out/r04_big_stats -o r04 -- $BIG
run r04_big_sq        --pmc $SQ -d $R/gpurun_out/r04_big_sq -o r04 -- $BIG
run r04_str_stats     --kernel-trace --stats -d $R/gpurun_out/r04_str_stats -o r04 -- $STR
run r04_str_sq        --pmc $SQ -d $R/gpurun_out/r04_str_sq -o r04 -- $STR
run r04_scan_stats    --kernel-trace --stats -d $R/gpurun_out/r04_scan_stats -o r04 -- $SCAN
run r04_scan_fetch    --pmc FETCH_SIZE -d $R/gpurun_out/r04_scan_fetch -o r04 -- $SCAN
run r04_scan_write    --pmc WRITE_SIZE -d $R/gpurun_out/r04_scan_write -o r04 -- $SCAN
run r04_scan_sq       --pmc $SQ -d $R/gpurun_out/r04_scan_sq -o r04 -- $SCAN
cd $R
P="python3 profiles/rocpd_summary.py"
db() { find gpurun_out/$1 -name '*.db' | head -1; }
$P stats $(db r04_stats) profiles/r04_kernel_stats.csv 3
$P pmc $(db r04_fetch) $(db r04_write) profiles/r04_pmc_traffic.json 64 1020000
$P sq $(db r04_sq) profiles/r04_sq_counters.csv
$P stats $(db r04_big_stats) profiles/r04_streams1024_kernel_stats.csv 3
$P sq $(db r04_big_sq) profiles/r04_streams1024_sq_counters.csv
$P stats $(db r04_str_stats) profiles/r04_stream_mode_kernel_stats.csv 3
$P sq $(db r04_str_sq) profiles/r04_stream_mode_sq_counters.csv
$P stats $(db r04_scan_stats) profiles/r04_scan12800_kernel_stats.csv 2
$P pmc $(db r04_scan_fetch) $(db r04_scan_write) profiles/r04_scan12800_pmc_traffic.json 12800 640000
$P sq $(db r04_scan_sq) profiles/r04_scan12800_sq_counters.csv
$P timeline $(db r04_scan_stats) profiles/r04_scan12800_timeline.csv 72
$P valu profiles/r04_valu_per_step.json calib_64=$(db r04_sq):1 calib_1024=$(db r04_big_sq):4 stream_mode_64=$(db r04_str_sq):1 scan_12800=$(db r04_scan_sq):s8
python3 tools/dist_cost.py > profiles/r04_dist_cost.json 2> gpurun_out/r04_dist_cost.err
tools/micro/clock_fp64 > profiles/r04_clock_lds_microbench.txt 2>&1
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* gpurun_out/profiles_r04/
# the raw databases stay on the box (gpurun copies back at most 64 MiB); the summaries above are what is kept
rm -rf gpurun_out/r04_stats gpurun_out/r04_fetch gpurun_out/r04_write gpurun_out/r04_sq gpurun_out/r04_big_stats gpurun_out/r04_big_sq gpurun_out/r04_str_stats gpurun_out/r04_str_sq gpurun_out/r04_scan_stats gpurun_out/r04_scan_fetch gpurun_out/r04_scan_write gpurun_out/r04_scan_sq
ls -la profiles/ | grep r04
tail -3 gpurun_out/r04_stats.log
