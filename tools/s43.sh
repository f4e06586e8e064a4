mkdir -p gpurun_out
O=gpurun_out/r04_s43.txt
: > $O
for rep in 1 2; do
for n in 400 800 1600; do
for st in 1 2 3 4; do
echo "captures $n stages $st: $(GSMCAL_SCAN_STAGES=$st python bench.py --workload scan --streams $n --frames 64 --distinct 32 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
")" >> $O
done; done; done
cat $O
