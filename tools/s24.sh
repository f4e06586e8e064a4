mkdir -p gpurun_out
O=gpurun_out/r04_s24.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/r04_s24.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print(d['ms_per_step'], d['roofline']['frac'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:4]))
"; }
for rep in 1 2; do
# scanner pipeline: nt on (default by size), inline in the pipeline or not, stage counts
for pipe in 0 1; do for st in 8 12 16 24 32; do
echo "scan 12800 nt pipe_inline $pipe stages $st: $(GSMCAL_SNR_INLINE_PIPE=$pipe GSMCAL_SCAN_STAGES=$st bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
done; done
echo "scan 12800 nt=0: $(GSMCAL_FRONT_NT=0 bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2 --no-kernel-events)" >> $O
for nt in 0 1; do
echo "scan 200 nt=$nt: $(GSMCAL_FRONT_NT=$nt bn --workload scan --streams 200 --frames 64 --distinct 32 --steps 50 --warmup 5)" >> $O
echo "calib 1024 nt=$nt: $(GSMCAL_FRONT_NT=$nt bn --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
echo "calib 64 nt=$nt: $(GSMCAL_FRONT_NT=$nt bn --steps 200 --warmup 20 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
done
for L in libgsmcal.so exp_ntst.so; do
echo "stream mode $L: $(GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L bn --mode stream --steps 20 --warmup 3 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
done
done
cat $O
