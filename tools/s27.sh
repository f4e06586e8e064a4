R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GSMCAL_SCAN_SPLIT=${SPLIT:-88}
python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --workload scan --streams 12800 --frames 64 --distinct 32 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_scan_tl -o r04 -- python3 $R/bench.py --no-cpu-baseline --no-sub --no-kernel-events --workload scan --streams 12800 --frames 64 --distinct 32 --steps 4 --warmup 2 > $R/gpurun_out/r04_scan_tl.log 2>&1
cd $R
python3 profiles/rocpd_summary.py timeline $(find gpurun_out/r04_scan_tl -name '*.db' | head -1) gpurun_out/r04_scan_timeline.csv 72
cat gpurun_out/r04_scan_timeline.csv
