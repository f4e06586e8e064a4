#!/bin/bash
# More randomised parity sweeps of the final round-4 code on other seeds (run on the GPU box): writes gpurun_out/r04_sweeps_extra.txt
mkdir -p gpurun_out
O=gpurun_out/r04_sweeps_extra.txt
{
echo "# Extra parity sweeps of the final round-4 code (other seeds than tools/sweeps_r04.sh)."
python tests/sweep_parity.py 4096 120000 102 64 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_parity.py 2048 130000 102 256 2>&1 | grep "sweep:\|MISMATCH\|status"      # one lane of 256: inline coarse SNRs, four-launch tail
python tests/sweep_parity.py 1024 140000 102 1024 2>&1 | grep "sweep:\|MISMATCH\|status"
python tests/sweep_scan.py 2600 15000 2>&1 | grep "sweep\|MISMATCH"                            # one call: eight pipeline stages, split front launches
python tests/sweep_scan.py 1400 19000 2>&1 | grep "sweep\|MISMATCH"                            # four stages
python tests/sweep_scan.py 300 21000 2>&1 | grep "sweep\|MISMATCH"                             # single stage, inline detector
} > $O
cat $O
