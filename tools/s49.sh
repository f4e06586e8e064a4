mkdir -p gpurun_out
O=gpurun_out/r04_s49.txt
: > $O
for rep in 1 2 3; do
for sp in 0 88; do
for n in 1024 2048; do
echo "split $sp streams $n: $(GSMCAL_SCAN_SPLIT=$sp python bench.py --streams $n --distinct 64 --steps 30 --warmup 3 --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/ab_streams.npy 2>>gpurun_out/r04_s49.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['config']['streams_calibrated_ok'])
")" >> $O
done; done; done
cat $O
