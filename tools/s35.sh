mkdir -p gpurun_out
O=gpurun_out/r04_s35.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s35.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2; do
for cfg in "1 24" "0 24" "0 16" "0 20" "1 20" "1 32" "0 32"; do
set -- $cfg
echo "inline_pipe $1 stages $2: $(GSMCAL_SNR_INLINE_PIPE=$1 GSMCAL_SCAN_STAGES=$2 bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done
cat $O
