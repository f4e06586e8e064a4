mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "stream or full_size" > gpurun_out/r04_s4_tests.log 2>&1; tail -2 gpurun_out/r04_s4_tests.log
for rep in 1 2; do
for L in multi-rtl-sdr-calibration_amd/lib/libgsmcal.so multi-rtl-sdr-calibration_amd/lib/exp_tile1016.so; do
GSMCAL_LIB=$PWD/$L python bench.py --no-sub --no-cpu-baseline --mode stream --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$L'.split('/')[-1], d['ms_per_step'], d.get('kernels_ms_per_step_untimed_pass'))
"
done; done > gpurun_out/r04_s4.txt 2>&1
cat gpurun_out/r04_s4.txt
