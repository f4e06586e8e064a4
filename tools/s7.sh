mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_s7_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s7_tests.log; tail -3 gpurun_out/r04_s7_tests.log
B="python bench.py --no-sub --no-cpu-baseline --cache-streams /tmp/s7_streams"
run() { "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d.get('kernels_ms_per_step_untimed_pass'))
"; }
: > gpurun_out/r04_s7.txt
for rep in 1 2; do
for L in libgsmcal.so exp_prev.so; do
echo "$L 64" >> gpurun_out/r04_s7.txt; GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L run $B --steps 200 --warmup 20 >> gpurun_out/r04_s7.txt
done; done
for L in libgsmcal.so exp_prev.so; do
echo "$L 1024" >> gpurun_out/r04_s7.txt; GSMCAL_LIB=$PWD/multi-rtl-sdr-calibration_amd/lib/$L run $B --steps 20 --warmup 3 --streams 1024 >> gpurun_out/r04_s7.txt
done
cat gpurun_out/r04_s7.txt
