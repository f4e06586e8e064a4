mkdir -p gpurun_out
O=gpurun_out/r04_s41.txt
: > $O
for rep in 1 2; do
for L in libgsmcal.so exp_f40.so exp_f20.so exp_f42.so exp_f6.so; do
echo "$L: $(GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L python bench.py --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/r04_s41.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'])
")" >> $O
done; done
cat $O
