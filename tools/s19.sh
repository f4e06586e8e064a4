mkdir -p gpurun_out
O=gpurun_out/r04_s19.txt
: > $O
bn() { python bench.py "$@" --no-cpu-baseline --no-kernel-events 2>>gpurun_out/r04_s19.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'])
"; }
for rep in 1 2; do
for m in 0 512; do
echo "scan 12800 inline_min $m: $(GSMCAL_SNR_INLINE_MIN=$m bn --workload scan --streams 12800 --frames 64 --distinct 32 --steps 10 --warmup 2)" >> $O
done
for m in 0 1; do
echo "scan 800 inline_min $m: $(GSMCAL_SNR_INLINE_MIN=$m bn --workload scan --streams 800 --frames 64 --distinct 32 --steps 30 --warmup 3)" >> $O
echo "scan 200 inline_min $m: $(GSMCAL_SNR_INLINE_MIN=$m bn --workload scan --streams 200 --frames 64 --distinct 32 --steps 50 --warmup 5)" >> $O
done
for m in 0 256; do
echo "calib 1024 inline_min $m: $(GSMCAL_SNR_INLINE_MIN=$m bn --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-sub --cache-streams /tmp/ab_streams.npy)" >> $O
done
done
cat $O
python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -k "config5 or config3" 2>&1 | tail -3
GSMCAL_SNR_INLINE_MIN=1 python tests/sweep_scan.py 1024 9000 2>&1 | grep "sweep\|MISMATCH" | tee -a $O
GSMCAL_SNR_INLINE_MIN=1 python tests/sweep_parity.py 1024 95000 2>&1 | grep "sweep:\|MISMATCH\|status" | tee -a $O
