mkdir -p gpurun_out
python tools/devtiming.py --workload scan --streams 534 --frames 64 --distinct 32 > gpurun_out/r04_s32_devtiming.txt 2>&1; grep -A1 "coarse_scan\|front" gpurun_out/r04_s32_devtiming.txt | cut -c1-600
