# A/B of the wave-role rotation in k_fine_cert: headline step + 1024 streams
mkdir -p gpurun_out
bash tools/abn.sh multi-rtl-sdr-calibration_amd/lib/libgsmcal.so multi-rtl-sdr-calibration_amd/lib/exp_rot.so
cp gpurun_out/abn.txt gpurun_out/r04_s16_abn.txt
: > gpurun_out/r04_s16_1024.txt
for rep in 1 2; do
for L in libgsmcal.so exp_rot.so; do
GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L python bench.py --streams 1024 --distinct 64 --steps 30 --warmup 3 --no-cpu-baseline --no-sub --cache-streams /tmp/ab_streams.npy 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        print('$L', d['ms_per_step'], ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in k.items()))
" >> gpurun_out/r04_s16_1024.txt
done; done
cat gpurun_out/r04_s16_1024.txt
