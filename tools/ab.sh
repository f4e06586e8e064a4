#!/bin/bash
# Development aid: A/B two builds of libgsmcal inside ONE gpurun call (box-to-box noise is ~1 us per step).
#   tools/ab.sh <libA.so> <libB.so> [bench args]   -> gpurun_out/ab.txt
mkdir -p gpurun_out
A=$1; B=$2; shift 2
: > gpurun_out/ab.txt
for rep in 1 2; do
  for L in "$A" "$B"; do
    echo "== $L $*" >> gpurun_out/ab.txt
    GSMCAL_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-sub --steps 200 --warmup 20 --cache-streams /tmp/ab_streams.npy "$@" 2>>gpurun_out/ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d.get('kernels_ms_per_step_untimed_pass'))
" >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt
