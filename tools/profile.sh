#!/bin/bash
# The round's profiles (run on the GPU box through gpurun: `tools/profile.sh r05`), ONE refresh for the round's kept code: rocprofv3 kernel stats (of the
# headline loop as it runs, calls in flight, and of the same loop fenced behind every step: the per-kernel figures of bench.py's event pass), HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) and one SQ pass, for the headline batch (64
# distinct streams), the 12 800-capture scanner batch, the 1 024-stream batch (throughput regime) and stream mode; plus the
# unprofiled default bench line, the N > 1 step cost on one rank (tools/dist_cost.py) and the clock / LDS micro-benchmark.
# Raw .db files land in gpurun_out/; profiles/rocpd_summary.py turns them into the small files kept under profiles/.
set -u
RT=${1:-r06}     # round tag: prefixes every file written under profiles/ and gpurun_out/
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GSMCAL_BENCH_NO_VARIANTS=1     # (the headline loop only: no depth-2 / one-buffer variants behind it)
B="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams"
CAL="$B --steps 20 --warmup 3"
CALC="$CAL --prewarm-steps 0"     # counter passes: the counters do not depend on the clock state, and 256 fewer steps keep the .db files small
BIG="$B --steps 10 --warmup 3 --streams 1024 --prewarm-steps 20"
STR="$B --steps 10 --warmup 3 --mode stream --prewarm-steps 20"
SCAN="$B --workload scan --streams 12800 --frames 64 --distinct 32 --steps 6 --warmup 2"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
$CAL > $R/gpurun_out/${RT}_bench_n1_noprof.json 2> /dev/null      # (fills the stream cache; also the unprofiled line)
run() { local name=$1; shift; rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1 || echo "rocprofv3 $name failed ($?)"; }
run ${RT}_stats         --kernel-trace --stats -d $R/gpurun_out/${RT}_stats -o ${RT} -- $CAL
run ${RT}_iso_stats     --kernel-trace --stats -d $R/gpurun_out/${RT}_iso_stats -o ${RT} -- $CAL --one-in-flight   # the same kernels, each with the GPU to itself
run ${RT}_fetch         --pmc FETCH_SIZE -d $R/gpurun_out/${RT}_fetch -o ${RT} -- $CALC
run ${RT}_write         --pmc WRITE_SIZE -d $R/gpurun_out/${RT}_write -o ${RT} -- $CALC
run ${RT}_sq            --pmc $SQ -d $R/gpurun_out/${RT}_sq -o ${RT} -- $CALC
run ${RT}_big_stats     --kernel-trace --stats -d $R/gpurun_out/${RT}_big_stats -o ${RT} -- $BIG
run ${RT}_big_sq        --pmc $SQ -d $R/gpurun_out/${RT}_big_sq -o ${RT} -- $BIG
run ${RT}_str_stats     --kernel-trace --stats -d $R/gpurun_out/${RT}_str_stats -o ${RT} -- $STR
run ${RT}_str_sq        --pmc $SQ -d $R/gpurun_out/${RT}_str_sq -o ${RT} -- $STR
run ${RT}_scan_stats    --kernel-trace --stats -d $R/gpurun_out/${RT}_scan_stats -o ${RT} -- $SCAN
run ${RT}_scan_fetch    --pmc FETCH_SIZE -d $R/gpurun_out/${RT}_scan_fetch -o ${RT} -- $SCAN
run ${RT}_scan_write    --pmc WRITE_SIZE -d $R/gpurun_out/${RT}_scan_write -o ${RT} -- $SCAN
run ${RT}_scan_sq       --pmc $SQ -d $R/gpurun_out/${RT}_scan_sq -o ${RT} -- $SCAN
cd $R
P="python3 profiles/rocpd_summary.py"
db() { find gpurun_out/$1 -name '*.db' | head -1; }
$P stats $(db ${RT}_stats) profiles/${RT}_kernel_stats.csv 3
$P stats $(db ${RT}_iso_stats) profiles/${RT}_kernel_stats_one_call_at_a_time.csv 3
$P pmc $(db ${RT}_fetch) $(db ${RT}_write) profiles/${RT}_pmc_traffic.json 64 1020000
$P sq $(db ${RT}_sq) profiles/${RT}_sq_counters.csv
$P stats $(db ${RT}_big_stats) profiles/${RT}_streams1024_kernel_stats.csv 3
$P sq $(db ${RT}_big_sq) profiles/${RT}_streams1024_sq_counters.csv
$P stats $(db ${RT}_str_stats) profiles/${RT}_stream_mode_kernel_stats.csv 3
$P sq $(db ${RT}_str_sq) profiles/${RT}_stream_mode_sq_counters.csv
$P stats $(db ${RT}_scan_stats) profiles/${RT}_scan12800_kernel_stats.csv 2
$P pmc $(db ${RT}_scan_fetch) $(db ${RT}_scan_write) profiles/${RT}_scan12800_pmc_traffic.json 12800 640000
$P sq $(db ${RT}_scan_sq) profiles/${RT}_scan12800_sq_counters.csv
$P timeline $(db ${RT}_scan_stats) profiles/${RT}_scan12800_timeline.csv 72
$P valu profiles/${RT}_valu_per_step.json calib_64=$(db ${RT}_sq):1 calib_1024=$(db ${RT}_big_sq):4 stream_mode_64=$(db ${RT}_str_sq):1 scan_12800=$(db ${RT}_scan_sq):s8
# which kernels these counters describe: the source hash (bench.py and the CPU suite refuse a summary of other sources) and the
# commit the tree was at when it was sent to the GPU box (written into profiles/.tree_commit before the gpurun call; the box has no .git)
python3 - profiles/${RT}_pmc_traffic.json profiles/${RT}_scan12800_pmc_traffic.json profiles/${RT}_valu_per_step.json <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import gsmcal
h = gsmcal.build.csrc_hash()
commit = open("profiles/.tree_commit").read().strip() if os.path.exists("profiles/.tree_commit") else "unknown"
for f in sys.argv[1:]:
    if os.path.exists(f):
        d = json.load(open(f))
        d["csrc_sha256"] = h
        d["git_commit"] = commit
        json.dump(d, open(f, "w"), indent=1)
PY
python3 tools/dist_cost.py > profiles/${RT}_dist_cost.json 2> gpurun_out/${RT}_dist_cost.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/clock_fp64.hip -o /tmp/clock_fp64 && /tmp/clock_fp64 > profiles/${RT}_clock_lds_microbench.txt 2>&1   # (built here: no binary in the tree)
mkdir -p gpurun_out/profiles_${RT} && cp profiles/${RT}_* gpurun_out/profiles_${RT}/
# the raw databases stay on the box (gpurun copies back at most 64 MiB); the summaries above are what is kept
rm -rf gpurun_out/${RT}_stats gpurun_out/${RT}_iso_stats gpurun_out/${RT}_fetch gpurun_out/${RT}_write gpurun_out/${RT}_sq gpurun_out/${RT}_big_stats gpurun_out/${RT}_big_sq gpurun_out/${RT}_str_stats gpurun_out/${RT}_str_sq gpurun_out/${RT}_scan_stats gpurun_out/${RT}_scan_fetch gpurun_out/${RT}_scan_write gpurun_out/${RT}_scan_sq
ls -la profiles/ | grep ${RT}_
tail -3 gpurun_out/${RT}_stats.log
