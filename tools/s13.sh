mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_s13_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s13_tests.log; tail -15 gpurun_out/r04_s13_tests.log
