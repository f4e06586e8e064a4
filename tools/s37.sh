mkdir -p gpurun_out
O=gpurun_out/r04_s37.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s37.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print(d['ms_per_step'], d['roofline']['frac'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:3]))
"; }
for rep in 1 2; do
for kb in 0 40 53 79; do
export GSMCAL_FRONT_LDS_KB=$kb
echo "lds $kb scan 12800: $(bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
echo "lds $kb scan 800: $(bn --workload scan --streams 800 --frames 64 --distinct 32 --steps 30 --warmup 3)" >> $O
echo "lds $kb calib 1024: $(bn --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-sub --cache-streams /tmp/ab_streams.npy --no-kernel-events)" >> $O
done; done
cat $O
