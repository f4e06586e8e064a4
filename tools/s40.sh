mkdir -p gpurun_out
O=gpurun_out/r04_s40.txt
: > $O
for rep in 1 2; do
for L in exp_p_fir0_rest2.so exp_p12.so exp_p4.so exp_p13.so; do
echo "$L: $(GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L python bench.py --mode stream --steps 20 --warmup 3 --no-cpu-baseline --no-sub --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/r04_s40.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print(d['ms_per_step'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:2]))
")" >> $O
done; done
cat $O
