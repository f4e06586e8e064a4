mkdir -p gpurun_out
O=gpurun_out/r04_s47.txt
: > $O
bn() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-sub --steps 200 --warmup 20 --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/r04_s47.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print(d['ms_per_step'], d['config']['streams_calibrated_ok'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:6]))
"; }
for rep in 1 2; do
echo "1 lane: $(bn)" >> $O
echo "2 lanes fused: $(GSMCAL_FUSE_MULTILANE=1 GSMCAL_LANES=2 GSMCAL_LANE_MIN=32 bn)" >> $O
echo "2 lanes unfused: $(GSMCAL_LANES=2 GSMCAL_LANE_MIN=32 bn)" >> $O
echo "4 lanes fused: $(GSMCAL_FUSE_MULTILANE=1 GSMCAL_LANES=4 GSMCAL_LANE_MIN=16 bn)" >> $O
echo "2 lanes fused staggered: $(GSMCAL_FUSE_MULTILANE=1 GSMCAL_LANES=2 GSMCAL_LANE_MIN=32 GSMCAL_LANE_STAGGER=1 bn)" >> $O
done
cat $O
