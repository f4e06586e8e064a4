mkdir -p gpurun_out
O=gpurun_out/r04_s30.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s30.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2; do
for sp in 0 70 80 88 94; do
echo "split $sp: $(GSMCAL_SCAN_SPLIT=$sp bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done
cat $O
GSMCAL_SCAN_SPLIT=85 python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -k "config5" 2>&1 | tail -2
