mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -k "bench_distributed" > gpurun_out/r04_s17_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s17_tests.log; tail -5 gpurun_out/r04_s17_tests.log
python tools/devtiming.py > gpurun_out/r04_s17_devtiming.txt 2>&1; tail -80 gpurun_out/r04_s17_devtiming.txt
