mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_s33_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_s33_tests.log; tail -3 gpurun_out/r04_s33_tests.log | head -2; grep -c . gpurun_out/r04_s33_tests.log; grep "passed\|failed" gpurun_out/r04_s33_tests.log
date +%s > /tmp/t0
bash tools/profile_r04.sh > gpurun_out/r04_s33_profile.log 2>&1; tail -25 gpurun_out/r04_s33_profile.log
echo "profile seconds: $(( $(date +%s) - $(cat /tmp/t0) ))"
