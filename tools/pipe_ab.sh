#!/bin/bash
# (Round-6 record: the staged forms -- stages 2 / 3 -- exist only with profiles/experiments_r06/staged_pipeline_and_side_fused.patch
# applied; the library in the tree has the side-by-side form, stages 1, and ignores GSMCAL_PIPE_STAGES.)
# A/B of the pipelined headline loop in ONE session on one box: depth (calls in flight) x stages (1: whole calls side by side,
# 2: front | tail, 3: front | fine search | fused tail).  Each run prints bench.py's line (headline + ms_per_step_depth1 / _llc_resident variants).
#   tools/pipe_ab.sh <tag> [depth:stages ...]   -> gpurun_out/<tag>/d<depth>_s<stages>.json
set -u
tag=${1:-pipe_ab}; shift
out=gpurun_out/$tag
mkdir -p "$out"
B="python bench.py --no-sub --no-cpu-baseline --no-kernel-events --cache-streams /tmp/gsmcal_streams"
if [ $# -eq 0 ]; then set -- 1:1 2:1 3:1 4:1 5:1 6:1 8:1 2:2 3:3 4:1; fi
for cfg in "$@"; do
  d=${cfg%%:*}; s=${cfg##*:}
  GSMCAL_PIPE_STAGES=$s $B --pipeline-depth "$d" >> "$out/d${d}_s${s}.json" 2>> "$out/d${d}_s${s}.err"
  echo "depth $d stages $s ${GSMCAL_PIPE_PRIO:-}: $(tail -1 "$out/d${d}_s${s}.json" | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print({k: r[k] for k in r if k.startswith("ms_per_step") and not k.endswith("what")}, r["tables_identical"])')"
done
