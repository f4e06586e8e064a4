#!/bin/bash
# Extra randomised parity sweeps (other seeds than tools/sweeps_r03.sh): writes gpurun_out/r03_sweeps_extra.txt
# usage: tools/sweeps_extra.sh [first_seed]   (default 90000)
mkdir -p gpurun_out
O=gpurun_out/r03_sweeps_extra.txt
F=${1:-90000}
{
python tests/sweep_parity.py 4096 $F 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 $((F + 10000)) 102 1024 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 1024 $((F + 20000)) 102 32 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_scan.py 2048 $((F / 10)) 2>&1 | grep "sweep\|MISMATCH"
} > $O
cat $O
