mkdir -p gpurun_out
O=gpurun_out/r04_s29.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s29.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2 3; do
for st in 17 18 20 22 24 28; do
echo "stages $st: $(GSMCAL_SCAN_STAGES=$st bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done
cat $O
