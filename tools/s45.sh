python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -k "bench_distributed" 2>&1 | tail -2
GSMCAL_FORCE_DIST=1 python bench.py --gpus 1 --no-sub --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['config']['collective'], d['config'].get('collective_autotune'), d['config']['gathered_table_checked_against_every_rank'])
"
