mkdir -p gpurun_out
: > gpurun_out/r04_s18.txt
for rep in 1 2; do
for ch in 0 1; do
for st in 16 12 20; do
echo "chain $ch stages $st" >> gpurun_out/r04_s18.txt
GSMCAL_SCAN_CHAIN=$ch GSMCAL_SCAN_STAGES=$st python bench.py --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events 2>>gpurun_out/r04_s18.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
" >> gpurun_out/r04_s18.txt
done; done; done
cat gpurun_out/r04_s18.txt
GSMCAL_SCAN_CHAIN=1 python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -k "config5 or config3" 2>&1 | tail -3
