#!/bin/bash
# ONE parametrised A/B session (replaces the per-session scripts tools/s1.sh ... s50.sh of round 4; they are in the git history,
# the table of what each compared is in profiles/NOTES_r04.md).  Box-to-box spread on the GPU pool is +-4 %, so variants are only
# ever compared inside ONE gpurun call:
#
#   gpurun --timeout 900 -- 'tools/ab_session.sh NAME REPS "label|ENV=1 ENV2=x|bench args" "label2||bench args" ...'
#
# Every variant runs `python bench.py <bench args> --no-cpu-baseline --no-sub` with its environment (GSMCAL_* switches,
# GSMCAL_LIB=<another build> for an A/B of two libraries), REPS times round-robin; one line per run goes to
# gpurun_out/NAME.txt: label, ms per step, calibrated streams, the six longest kernels of the untimed event pass.
# Empty bench args default to the headline shape with 200 timed steps.
NAME=$1; REPS=$2; shift 2
mkdir -p gpurun_out
O=gpurun_out/$NAME.txt
: > "$O"
for rep in $(seq 1 "$REPS"); do
  for v in "$@"; do
    IFS='|' read -r label envs bargs <<< "$v"
    [ -z "$bargs" ] && bargs="--steps 200 --warmup 20"
    line=$(env $envs timeout 600 python bench.py $bargs --no-cpu-baseline --no-sub --cache-streams /tmp/ab_streams.npy 2>>"gpurun_out/$NAME.err" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels_ms_per_step_untimed_pass') or {}
        ks = ' '.join('%s=%.1f' % (a.strip('()').split('<')[0][2:], 1e3 * b) for a, b in list(k.items())[:6] if isinstance(b, float))
        print(d['ms_per_step'], d.get('ms_per_step_no_prewarm'), d.get('config', {}).get('streams_calibrated_ok'), d['roofline']['frac'], ks)
")
    echo "$label: $line" >> "$O"
  done
done
cat "$O"
