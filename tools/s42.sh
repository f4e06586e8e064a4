mkdir -p gpurun_out
O=gpurun_out/r04_s42.txt
: > $O
for rep in 1 2; do
for L in libgsmcal.so exp_s12.so exp_s13.so exp_s14.so exp_s7.so; do
echo "$L: $(GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L python bench.py --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events 2>>gpurun_out/r04_s42.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
")" >> $O
done; done
cat $O
