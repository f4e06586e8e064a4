#!/bin/bash
# Development aid: instruction-fetch counters of the 64-stream step (is the post-chain kernel waiting for its own code?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CAL="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams --steps 20 --warmup 3"
$CAL > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $R/gpurun_out/ic1 -o ic -- $CAL > $R/gpurun_out/ic1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/ic2 -o ic -- $CAL > $R/gpurun_out/ic2.log 2>&1
cd $R
for d in ic1 ic2; do python3 profiles/rocpd_summary.py sq $(find gpurun_out/$d -name '*.db' | head -1) gpurun_out/$d.csv; cat gpurun_out/$d.csv; done
