#!/bin/bash
# same-session A/B of stream mode (18 B/sample): the library in the tree against other builds (GSMCAL_LIB)
#   tools/stream_ab.sh [lib.so ...]
B="python bench.py --mode stream --no-sub --no-cpu-baseline --no-kernel-events --steps 10 --raw-buffers 1 --cache-streams /tmp/gs"
$B > /dev/null 2>&1
for rep in 1 2; do
  for lib in "" "$@"; do
    echo -n "${lib:-tree}: "
    env ${lib:+GSMCAL_LIB=$lib} $B 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["roofline"]["frac"])'
  done
done
