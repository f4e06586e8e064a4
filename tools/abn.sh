#!/bin/bash
# Development aid: several builds of libgsmcal in one GPU session: tools/abn.sh lib1.so lib2.so ... (step + kernel times)
mkdir -p gpurun_out; : > gpurun_out/abn.txt
for rep in 1 2; do
  for L in "$@"; do
    GSMCAL_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-sub --steps 200 --warmup 20 --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/abn.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print('$L'.split('/')[-1], d['ms_per_step'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in k.items()))
" >> gpurun_out/abn.txt
  done
done
cat gpurun_out/abn.txt
