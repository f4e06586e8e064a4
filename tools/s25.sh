mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_s25_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s25_tests.log; tail -4 gpurun_out/r04_s25_tests.log
O=gpurun_out/r04_s25.txt
: > $O
python tests/sweep_scan.py 2048 11000 2>&1 | grep "sweep\|MISMATCH" | tee -a $O
python tests/sweep_scan.py 700 13100 2>&1 | grep "sweep\|MISMATCH" | tee -a $O
python tests/sweep_parity.py 1024 97000 2>&1 | grep "sweep:\|MISMATCH\|status" | tee -a $O
python tests/sweep_parity.py 64 98000 2>&1 | grep "sweep:\|MISMATCH\|status" | tee -a $O
python bench.py > gpurun_out/r04_s25_bench.json 2> gpurun_out/r04_s25_bench.err; echo rc $?
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/r04_s25_bench.json') if l.startswith('{')][-1]
print(d['ms_per_step'], d['value'])
for k,v in d['sub_results'].items():
    print(k, {a:b for a,b in v.items() if a in ('ms_per_step','ms_per_call','path_frac_of_hbm','error','tables_identical_to_headline')})
PY
