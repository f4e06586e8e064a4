mkdir -p gpurun_out
T0=$(date +%s)
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err; echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('roofline_compute'))
for k,v in d['sub_results'].items():
    print(k, {a:b for a,b in v.items() if a in ('ms_per_step','ms_per_call','path_frac_of_hbm','error','collective','roofline_compute','sustained_GBps_with_detector','stderr_tail','gathered_table_checked_against_every_rank')})
print(d.get('cpu_baseline'))
PY
