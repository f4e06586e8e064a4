mkdir -p gpurun_out
python bench.py > gpurun_out/r04_s22_bench.json 2> gpurun_out/r04_s22_bench.err; echo rc $?
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/r04_s22_bench.json') if l.startswith('{')][-1]
print(d['ms_per_step'], d['value'])
for k,v in d['sub_results'].items():
    print(k, {a:b for a,b in v.items() if a in ('ms_per_step','ms_per_call','path_frac_of_hbm','error','tables_identical_to_headline','kernels_ms_per_step_untimed_pass')})
PY
