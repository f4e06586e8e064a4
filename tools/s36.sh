mkdir -p gpurun_out
O=gpurun_out/r04_s36.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s36.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print(d['ms_per_step'], d['roofline']['frac'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:3]))
"; }
for rep in 1 2; do
for L in libgsmcal.so exp_full.so; do
export GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L
echo "$L scan 12800: $(bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
echo "$L scan 800: $(bn --workload scan --streams 800 --frames 64 --distinct 32 --steps 30 --warmup 3)" >> $O
echo "$L scan 200: $(bn --workload scan --streams 200 --frames 64 --distinct 32 --steps 50 --warmup 5)" >> $O
echo "$L calib 64: $(bn --steps 200 --warmup 20 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
echo "$L calib 1024: $(bn --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
done; done
cat $O
unset GSMCAL_LIB
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "front or raw2iq or config5 or pipeline_choices or filter" 2>&1 | tail -3
