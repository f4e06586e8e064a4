mkdir -p gpurun_out
O=gpurun_out/r04_s50.txt
: > $O
for rep in 1 2; do
for kb in 0 54 64 81; do
for st in 20 26; do
echo "det lds $kb stages $st: $(GSMCAL_DET_LDS_KB=$kb GSMCAL_SCAN_STAGES=$st python bench.py --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
")" >> $O
done; done; done
cat $O
