python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider 2>&1 | tail -2
for n in 1300 1900; do python bench.py --workload scan --streams $n --frames 64 --distinct 32 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print($n, d['ms_per_step'], d['roofline']['frac'], d.get('parity_checked_captures'))
"; done
