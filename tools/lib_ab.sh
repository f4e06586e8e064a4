#!/bin/bash
# same-session A/B of library builds on the headline (64 streams, rotated input), the 1 024-stream batch and the mixed batch:
#   tools/lib_ab.sh [lib.so ...]      ("" = the library in the tree)
B="python bench.py --no-sub --no-cpu-baseline --no-kernel-events --cache-streams /tmp/gs"
GSMCAL_BENCH_NO_VARIANTS=1 $B > /dev/null 2>&1
for rep in 1 2; do
  for lib in "" "$@"; do
    h=$(env ${lib:+GSMCAL_LIB=$lib} GSMCAL_BENCH_NO_VARIANTS=1 $B 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["config"]["streams_calibrated_ok"])')
    b=$(env ${lib:+GSMCAL_LIB=$lib} GSMCAL_BENCH_NO_VARIANTS=1 $B --streams 1024 --steps 10 --prewarm-steps 20 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"])')
    echo "$(basename ${lib:-tree}): 64 streams $h | 1024 streams $b"
  done
done
