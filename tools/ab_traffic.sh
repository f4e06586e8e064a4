#!/bin/bash
# the 1.72x whole-step traffic of the 64-stream headline, each contributor removed in turn -- step time (bench.py's
# line without the profiler) and HBM bytes per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over
# the chain's kernels as bench.py's roofline.traffic does) before / after, ONE session on one box.
#   tools/ab_traffic.sh <tag>  ->  profiles/<tag>_ab_traffic.txt
# Variants:
#   default        the product library
#   snr_moving     GSMCAL_SNR_FULL=0: the SNR table holds the moving search's 3 579 windows only, the hop walk computes its own spectra
#   no_reuse       -DGSMCAL_AB_NO_REUSE_L0: per-burst gathers filter their raw bytes again instead of reading the fine windows' samples
#   lazy_window    -DGSMCAL_AB_LAZY_WIN: k_fine_cert writes a window out only when its certificate left chunks open (+ no_reuse)
#   lazy+moving    both
# (the two -D builds are made by: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSMCAL_AB_<X> csrc/gsmcal.hip -o lib/libgsmcal_ab_<X>.so -ldl)
set -u
RT=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/multi-rtl-sdr-calibration_amd/lib
OUT=$R/profiles/${RT}_ab_traffic.txt
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --cache-streams /tmp/gsmcal_streams --steps 20 --warmup 3"
$B > /dev/null 2>&1      # (fills the stream cache)
: > $OUT
echo "# tools/ab_traffic.sh $RT: 64 distinct streams x 1 020 000, table mode, raw batch rotated over 4 buffers; algorithmic 130.56 MB per step" >> $OUT
echo "# variant | ms_per_step (two runs) | HBM MB per step (PMC) | traffic / algorithmic | per kernel MB" >> $OUT
run() {
  local name=$1 lib=$2 env=$3
  local t1 t2
  t1=$(env $env ${lib:+GSMCAL_LIB=$lib} $B 2>/dev/null | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["tables_identical"], r["config"]["streams_calibrated_ok"])')
  t2=$(env $env ${lib:+GSMCAL_LIB=$lib} $B 2>/dev/null | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"])')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/ab_$c
    env $env ${lib:+GSMCAL_LIB=$lib} GSMCAL_BENCH_NO_VARIANTS=1 rocprofv3 --pmc $c -d $R/gpurun_out/ab_$c -o ab -- $B --prewarm-steps 0 > $R/gpurun_out/ab_${name}_$c.log 2>&1
  done
  python3 $R/profiles/rocpd_summary.py pmc $(find $R/gpurun_out/ab_FETCH_SIZE -name '*.db' | head -1) $(find $R/gpurun_out/ab_WRITE_SIZE -name '*.db' | head -1) $R/gpurun_out/ab_${name}_pmc.json 64 1020000 > /dev/null 2>&1
  python3 - "$name" "$t1" "$t2" $R/gpurun_out/ab_${name}_pmc.json >> $OUT <<'PY'
import json, sys
name, t1, t2, f = sys.argv[1:5]
p = json.load(open(f))["hbm_bytes_per_launch"]
per = {k.split("<")[0]: v for k, v in p.items() if k.startswith("k_") and k != "k_make_twiddles"}
tot = sum(per.values())
print(f"{name:12s} | {t1} / {t2} | {tot / 1e6:7.1f} | {tot / 130.56e6:5.2f} | " + " ".join(f"{k}={v / 1e6:.1f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])))
PY
  rm -rf $R/gpurun_out/ab_FETCH_SIZE $R/gpurun_out/ab_WRITE_SIZE
}
run default      ""                              "A=1"
run snr_moving   ""                              "GSMCAL_SNR_FULL=0"
run no_reuse     $L/libgsmcal_ab_NO_REUSE_L0.so  "A=1"
run lazy_window  $L/libgsmcal_ab_LAZY_WIN.so     "A=1"
run lazy+moving  $L/libgsmcal_ab_LAZY_WIN.so     "GSMCAL_SNR_FULL=0"
run default      ""                              "A=1"
cat $OUT
