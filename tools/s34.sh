mkdir -p gpurun_out
S=$(date +%s)
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err; echo "rc $? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/r04_bench_n1.json') if l.startswith('{')][-1]
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('cpu_baseline'))
print(d.get('kernel_roofline'))
print(d.get('roofline_compute'))
for k,v in d['sub_results'].items():
    print(k, {a:b for a,b in v.items() if a in ('ms_per_step','ms_per_call','path_frac_of_hbm','error','tables_identical_to_headline','roofline_compute','GBps','sustained_GBps')})
PY
