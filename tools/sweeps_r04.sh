#!/bin/bash
# Randomised parity sweeps against the oracle (run on the GPU box): writes gpurun_out/r04_sweeps.txt
mkdir -p gpurun_out
O=gpurun_out/r04_sweeps.txt
{
echo "# Randomised parity sweeps of the final round-4 code against the oracle (tests/sweep_parity.py N first_dongle [frames] [batch]; tests/sweep_scan.py)."
echo "# 64-stream batches run the fused k_post_chain_r tail, the 2048-stream batch four lanes of the four-launch tail."
python tests/sweep_parity.py 2048 50000 2>&1 | grep "sweep:\|MISMATCH\|status"
for f in 60000 61100 61200 61300; do python tests/sweep_parity.py 64 $f 2>&1 | grep "sweep:\|MISMATCH\|status"; done
python tests/sweep_scan.py 2048 7000 2>&1 | grep "sweep\|MISMATCH"
python tests/sweep_parity.py 4096 70000 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 512 80000 102 48 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 90000 2>&1 | grep "sweep:\|MISMATCH\|status"     # one 1024-stream batch: four staggered lanes
} > $O
cat $O
