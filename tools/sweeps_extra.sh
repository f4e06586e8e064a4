#!/bin/bash
# Extra randomised parity sweeps (other seeds than tools/sweeps_r03.sh): appends to gpurun_out/r03_sweeps_extra.txt
mkdir -p gpurun_out
O=gpurun_out/r03_sweeps_extra.txt
{
python tests/sweep_parity.py 4096 90000 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 100000 102 1024 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 110000 102 32 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_scan.py 2048 9000 2>&1 | grep "sweep\|MISMATCH"
} > $O
cat $O
