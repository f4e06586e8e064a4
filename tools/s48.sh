for v in "" "GSMCAL_FUSE_POST=0" "GSMCAL_POST_REPL=0"; do
echo "$v: $(env $v python bench.py --no-cpu-baseline --no-sub --steps 200 --warmup 20 --no-kernel-events --cache-streams /tmp/ab_streams.npy 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'])
")"
done
