mkdir -p gpurun_out
B="python bench.py --no-sub --no-cpu-baseline --no-kernel-events --cache-streams /tmp/s9_streams --steps 20 --warmup 3 --streams 1024"
run() { "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'])
"; }
: > gpurun_out/r04_s9.txt
for cfg in "" "GSMCAL_LANE_STAGGER=1" "GSMCAL_LANES=2" "GSMCAL_LANES=3" "GSMCAL_LANES=6 GSMCAL_LANE_STAGGER=1" "GSMCAL_LANES=8 GSMCAL_LANE_STAGGER=1" "GSMCAL_LANES=8 GSMCAL_LANE_MIN=128" "GSMCAL_GRAPH=0"; do
echo "cfg [$cfg]" >> gpurun_out/r04_s9.txt; env $cfg $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'])
" >> gpurun_out/r04_s9.txt
done
cat gpurun_out/r04_s9.txt
