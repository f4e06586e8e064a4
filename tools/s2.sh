mkdir -p gpurun_out
python tools/dist_cost.py --variants ADE > gpurun_out/r04_s2_dist_relDev.json 2> gpurun_out/r04_s2_dist1.err
GSMCAL_AG_EVENT_FLAGS=2 python tools/dist_cost.py --variants ADE > gpurun_out/r04_s2_dist_flags2.json 2> gpurun_out/r04_s2_dist2.err
python bench.py --no-sub --no-cpu-baseline --no-kernel-events > gpurun_out/r04_s2_bench_irq.json 2> /dev/null
HSA_ENABLE_INTERRUPT=0 python bench.py --no-sub --no-cpu-baseline --no-kernel-events > gpurun_out/r04_s2_bench_noirq.json 2> /dev/null
HSA_ENABLE_INTERRUPT=0 python tools/dist_cost.py --variants ADE > gpurun_out/r04_s2_dist_noirq.json 2> /dev/null
cat gpurun_out/r04_s2_dist_relDev.json gpurun_out/r04_s2_dist_flags2.json gpurun_out/r04_s2_dist_noirq.json
python - <<'PY'
import json
for f in ('irq','noirq'):
    d=json.loads(open(f'gpurun_out/r04_s2_bench_{f}.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'])
PY
