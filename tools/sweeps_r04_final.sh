#!/bin/bash
# Parity sweeps of the round's LAST commit on fresh seeds (run on the GPU box): writes gpurun_out/r04_sweeps_final.txt
mkdir -p gpurun_out
O=gpurun_out/r04_sweeps_final.txt
{
echo "# Parity sweeps of the final round-4 build on seeds no earlier sweep used."
python tests/sweep_parity.py 4096 200000 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 210000 102 1024 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 768 220000 102 384 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 520 230000 61 130 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_scan.py 3000 25000 2>&1 | grep "sweep\|MISMATCH"
python tests/sweep_scan.py 1250 29000 2>&1 | grep "sweep\|MISMATCH"
python tests/sweep_scan.py 640 31000 2>&1 | grep "sweep\|MISMATCH"
} > $O
cat $O
