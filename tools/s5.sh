mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_worstcase.py -q -p no:cacheprovider -x > gpurun_out/r04_s5_tests.log 2>&1; tail -2 gpurun_out/r04_s5_tests.log
B="python bench.py --no-sub --no-cpu-baseline --cache-streams /tmp/s5_streams"
run() { "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d.get('kernels_ms_per_step_untimed_pass'))
"; }
for rep in 1 2; do
echo "fcert_s47=1 64"; run $B --steps 200 --warmup 20
echo "fcert_s47=0 64"; GSMCAL_FCERT_S47=0 run $B --steps 200 --warmup 20
done > gpurun_out/r04_s5.txt 2>&1
echo "fcert_s47=1 1024" >> gpurun_out/r04_s5.txt; run $B --steps 20 --warmup 3 --streams 1024 >> gpurun_out/r04_s5.txt
echo "fcert_s47=0 1024" >> gpurun_out/r04_s5.txt; GSMCAL_FCERT_S47=0 run $B --steps 20 --warmup 3 --streams 1024 >> gpurun_out/r04_s5.txt
echo "stream mode" >> gpurun_out/r04_s5.txt; run $B --steps 10 --mode stream >> gpurun_out/r04_s5.txt
cat gpurun_out/r04_s5.txt
