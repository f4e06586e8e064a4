mkdir -p gpurun_out
for i in 1 2 3; do python bench.py --no-sub --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['prewarm_steps'], d['roofline'].get('frac'), d.get('kernel_roofline', {}) )
"; done
python bench.py --no-sub --no-cpu-baseline --prewarm-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('no prewarm', d['ms_per_step'])
"
