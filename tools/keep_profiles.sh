#!/bin/bash
# After `gpurun -- tools/profile.sh <tag>`: copy the counter / stats summaries the box produced (gpurun_out/profiles_<tag>/) into
# profiles/, leaving alone the files of the round that were NOT made by profile.sh (sweeps, bench record, A/B tables, timelines).
tag=${1:-r06}
for f in gpurun_out/profiles_$tag/${tag}_*; do
  b=$(basename $f)
  case $b in ${tag}_sweeps*|${tag}_mfma_f64_overlap.txt|${tag}_bench_n1.json|${tag}_ab_traffic.txt|${tag}_pipe_*|${tag}_console_example.txt) ;; *) cp $f profiles/$b;; esac
done
cp gpurun_out/${tag}_bench_n1_noprof.json profiles/${tag}_bench_n1_noprof.json 2>/dev/null
python3 -c "
import json, sys
sys.path.insert(0, '.')
import gsmcal
d = json.load(open('profiles/${tag}_pmc_traffic.json'))
print('profile hash', d.get('csrc_sha256', '?')[:12], 'tree hash', gsmcal.build.csrc_hash()[:12], 'commit', d.get('git_commit'))"
