#!/bin/bash
# Development aid: 1024-stream step against the number of lanes (GSMCAL_LANES) and the fewest streams per lane.
mkdir -p gpurun_out; : > gpurun_out/lanes.txt
for L in 1 2 4 8 16; do
  for M in 64; do
    echo "== lanes $L lane_min $M" >> gpurun_out/lanes.txt
    GSMCAL_LANES=$L GSMCAL_LANE_MIN=$M python bench.py --no-cpu-baseline --no-sub --no-kernel-events --streams 1024 --steps 20 --warmup 5 --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'])
" >> gpurun_out/lanes.txt
  done
done
cat gpurun_out/lanes.txt
