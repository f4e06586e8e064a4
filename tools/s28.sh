mkdir -p gpurun_out
O=gpurun_out/r04_s28.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s28.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2; do
for gr in 1 0; do for ord in 0 1 2; do
echo "graph $gr order $ord: $(GSMCAL_GRAPH=$gr GSMCAL_SCAN_ORDER=$ord bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done; done
cat $O
