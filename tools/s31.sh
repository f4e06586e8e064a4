mkdir -p gpurun_out
O=gpurun_out/r04_s31.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s31.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2; do
for cfg in "1 0 24" "1 88 24" "2 0 24" "2 0 32" "2 88 24" "2 0 16" "3 0 32"; do
set -- $cfg
echo "dep $1 split $2 stages $3: $(GSMCAL_SCAN_DEP=$1 GSMCAL_SCAN_SPLIT=$2 GSMCAL_SCAN_STAGES=$3 bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done
cat $O
