#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes, as MI355X_MICROARCH.md prescribes) and one SQ pass, for the headline batch and the 12 800-capture scanner
# batch.  Raw .db files land in gpurun_out/; profiles/rocpd_summary.py turns them into the small files kept under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events"
CAL="$B --steps 20 --warmup 3"
SCAN="$B --workload scan --streams 12800 --frames 64 --distinct 32 --steps 6 --warmup 2"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
run() { local name=$1; shift; rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1 || echo "rocprofv3 $name failed ($?)"; }
run r02_stats       --kernel-trace --stats -d $R/gpurun_out/r02_stats -o r02 -- $CAL
run r02_fetch       --pmc FETCH_SIZE -d $R/gpurun_out/r02_fetch -o r02 -- $CAL
run r02_write       --pmc WRITE_SIZE -d $R/gpurun_out/r02_write -o r02 -- $CAL
run r02_sq          --pmc $SQ -d $R/gpurun_out/r02_sq -o r02 -- $CAL
run r02_scan_stats  --kernel-trace --stats -d $R/gpurun_out/r02_scan_stats -o r02 -- $SCAN
run r02_scan_fetch  --pmc FETCH_SIZE -d $R/gpurun_out/r02_scan_fetch -o r02 -- $SCAN
run r02_scan_write  --pmc WRITE_SIZE -d $R/gpurun_out/r02_scan_write -o r02 -- $SCAN
run r02_scan_sq     --pmc $SQ -d $R/gpurun_out/r02_scan_sq -o r02 -- $SCAN
cd $R
P="python3 profiles/rocpd_summary.py"
db() { find gpurun_out/$1 -name '*.db' | head -1; }
$P stats $(db r02_stats) profiles/r02_kernel_stats.csv 3
$P pmc $(db r02_fetch) $(db r02_write) profiles/r02_pmc_traffic.json 64 1020000
$P sq $(db r02_sq) profiles/r02_sq_counters.csv
$P stats $(db r02_scan_stats) profiles/r02_scan12800_kernel_stats.csv 2
$P pmc $(db r02_scan_fetch) $(db r02_scan_write) profiles/r02_scan12800_pmc_traffic.json 12800 640000
$P sq $(db r02_scan_sq) profiles/r02_scan12800_sq_counters.csv
mkdir -p gpurun_out/profiles_r02 && cp profiles/r02_* gpurun_out/profiles_r02/
ls -la profiles/ | tail -12
tail -3 gpurun_out/r02_stats.log
