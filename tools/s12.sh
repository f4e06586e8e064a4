mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "stream or full_size" > gpurun_out/r04_s12_tests.log 2>&1; tail -2 gpurun_out/r04_s12_tests.log
: > gpurun_out/r04_s12.txt
for rep in 1 2; do
for L in libgsmcal.so exp_prev.so; do
GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L python bench.py --no-sub --no-cpu-baseline --mode stream --steps 10 --cache-streams /tmp/s12 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$L', d['ms_per_step'], {k:v for k,v in d.get('kernels_ms_per_step_untimed_pass').items() if 'stream' in k})
" >> gpurun_out/r04_s12.txt
done; done
cat gpurun_out/r04_s12.txt
