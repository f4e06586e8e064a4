mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_s1_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_s1_tests.log
python bench.py --no-sub --no-cpu-baseline > gpurun_out/r04_s1_bench.json 2> gpurun_out/r04_s1_bench.err
python tools/dist_cost.py > gpurun_out/r04_s1_dist.json 2> gpurun_out/r04_s1_dist.err
GSMCAL_FORCE_DIST=1 python bench.py --no-sub --no-cpu-baseline --no-kernel-events > gpurun_out/r04_s1_forcedist.json 2> gpurun_out/r04_s1_forcedist.err
tail -5 gpurun_out/r04_s1_tests.log; cat gpurun_out/r04_s1_dist.json; tail -3 gpurun_out/r04_s1_dist.err
