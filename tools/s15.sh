mkdir -p gpurun_out
: > gpurun_out/r04_s15.txt
for n in 200 800; do
for st in 1 2 3 4 6; do
echo "captures $n stages $st" >> gpurun_out/r04_s15.txt
GSMCAL_SCAN_STAGES=$st python bench.py --workload scan --streams $n --frames 64 --distinct 32 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
" >> gpurun_out/r04_s15.txt
done; done
cat gpurun_out/r04_s15.txt
