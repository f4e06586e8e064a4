#!/bin/bash
# Development aid: 12 800-capture scanner step against the number of pipeline stages (GSMCAL_SCAN_STAGES)
mkdir -p gpurun_out; : > gpurun_out/stages.txt
for L in 8 12 16 20 24 32; do
  echo "== stages $L" >> gpurun_out/stages.txt
  GSMCAL_SCAN_STAGES=$L python bench.py --no-cpu-baseline --no-sub --no-kernel-events --workload scan --streams 12800 --frames 64 --distinct 32 --steps 6 --warmup 2 2>>gpurun_out/stages.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'])
" >> gpurun_out/stages.txt
done
cat gpurun_out/stages.txt
