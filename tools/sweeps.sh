#!/bin/bash
# Randomised parity sweeps of the current build against the oracle, on the GPU box: tools/sweeps.sh TAG FIRST_SEED
#   -> gpurun_out/TAG_sweeps.txt (copy into profiles/ to keep).  Fresh FIRST_SEED per round: no sweep repeats another's streams.
# tests/sweep_parity.py N first_dongle [frames] [batch]; tests/sweep_scan.py N first_unit.  Batches of 64 run the fused tail,
# 130 / 384 two and four lanes of the four-launch tail, 1 024 and 2 048 the staggered throughput lanes with the inline detector.
TAG=${1:-r05}; S=${2:-300000}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_sweeps.txt
{
echo "# Parity sweeps of the $TAG build against the oracle, seeds from $S."
python tests/sweep_parity.py 2048 $S 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 $((S + 10000)) 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 $((S + 20000)) 102 1024 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 768 $((S + 30000)) 102 384 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 520 $((S + 40000)) 61 130 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_scan.py 3000 $((S + 50000)) 2>&1 | grep "sweep\|MISMATCH"
python tests/sweep_scan.py 1250 $((S + 60000)) 2>&1 | grep "sweep\|MISMATCH"
python tests/sweep_scan.py 640 $((S + 70000)) 2>&1 | grep "sweep\|MISMATCH"
} > $O
cat $O
