mkdir -p gpurun_out
: > gpurun_out/r04_s10.txt
for n in 128 256 512 1024 2048; do
for cfg in "GSMCAL_LANE_STAGGER=0" "GSMCAL_LANE_STAGGER=1"; do
echo "streams $n [$cfg]" >> gpurun_out/r04_s10.txt; env $cfg python bench.py --no-sub --no-cpu-baseline --no-kernel-events --cache-streams /tmp/s10_streams --steps 20 --warmup 3 --streams $n 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'])
" >> gpurun_out/r04_s10.txt
done; done
cat gpurun_out/r04_s10.txt
