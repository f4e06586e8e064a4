mkdir -p gpurun_out
O=gpurun_out/r04_s46.txt
: > $O
for rep in 1 2 3; do
for tp in 0 25 40 60; do
echo "taper $tp: $(GSMCAL_SCAN_TAPER=$tp python bench.py --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
")" >> $O
done; done
cat $O
