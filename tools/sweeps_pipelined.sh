#!/bin/bash
# Randomised parity sweeps of the CALLS-IN-FLIGHT form (gsmcal_ctx_set_pipeline_depth; what bench.py's headline and its 200-capture
# scanner line run): every batch against the oracle one call at a time, then the same batches DEPTH calls in flight, bit for bit.
#   tools/sweeps_pipelined.sh TAG FIRST_SEED  -> gpurun_out/TAG_sweeps_pipelined.txt
TAG=${1:-r06}; S=${2:-1300000}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_sweeps_pipelined.txt
{
echo "# Parity sweeps of the $TAG build with calls in flight, seeds from $S."
python tests/sweep_parity.py 2048 $S 61 64 4 2>&1 | grep "sweep:\|MISMATCH\|status\|differ"
python tests/sweep_parity.py 1024 $((S + 10000)) 102 64 3 2>&1 | grep "sweep:\|MISMATCH\|status\|differ"
python tests/sweep_parity.py 512 $((S + 20000)) 61 32 8 2>&1 | grep "sweep:\|MISMATCH\|status\|differ"
python tests/sweep_parity.py 512 $((S + 30000)) 61 64 2 2>&1 | grep "sweep:\|MISMATCH\|status\|differ"
python tests/sweep_scan.py 3000 $((S + 50000)) 200 4 2>&1 | grep "sweep\|mismatch"
python tests/sweep_scan.py 1200 $((S + 60000)) 48 8 2>&1 | grep "sweep\|mismatch"
} > $O
cat $O
