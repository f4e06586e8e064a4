#!/bin/bash
# (Round-6 record: stages 2 / 3 need profiles/experiments_r06/staged_pipeline_and_side_fused.patch; the library in the tree runs stages 1 only.)
# rocprofv3 kernel timeline of the pipelined headline loop: who overlaps whom.
#   tools/pipe_timeline.sh <tag> <depth> <stages>  -> gpurun_out/<tag>_timeline_d<depth>_s<stages>.csv (last 72 dispatches)
set -u
tag=${1:-pt}; depth=${2:-2}; stages=${3:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GSMCAL_PIPE_STAGES=$stages GSMCAL_BENCH_NO_VARIANTS=1
python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams --steps 20 --warmup 3 --prewarm-steps 20 --pipeline-depth $depth > /dev/null 2>&1
rm -rf $R/gpurun_out/${tag}_db
rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_db -o pt -- python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams --steps 20 --warmup 3 --prewarm-steps 20 --pipeline-depth $depth > $R/gpurun_out/${tag}_d${depth}_s${stages}.log 2>&1
cd $R
python3 profiles/rocpd_summary.py timeline $(find gpurun_out/${tag}_db -name '*.db' | head -1) gpurun_out/${tag}_timeline_d${depth}_s${stages}.csv 72
rm -rf gpurun_out/${tag}_db
grep '"metric"' gpurun_out/${tag}_d${depth}_s${stages}.log | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["pipeline_depth"])'
