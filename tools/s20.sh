mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_s20_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s20_tests.log; tail -8 gpurun_out/r04_s20_tests.log
