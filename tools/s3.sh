mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "stream or full_size or golden" > gpurun_out/r04_s3_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_s3_tests.log
python bench.py --no-sub --no-cpu-baseline --no-kernel-events --mode stream --steps 10 > gpurun_out/r04_s3_stream_new.json 2> /dev/null
GSMCAL_STREAM_S47=0 python bench.py --no-sub --no-cpu-baseline --no-kernel-events --mode stream --steps 10 > gpurun_out/r04_s3_stream_old.json 2> /dev/null
python bench.py --no-sub --no-cpu-baseline --mode stream --steps 10 > gpurun_out/r04_s3_stream_new_ev.json 2> /dev/null
tail -3 gpurun_out/r04_s3_tests.log
python - <<'PY'
import json
for f in ('new','old','new_ev'):
    d=json.loads(open(f'gpurun_out/r04_s3_stream_{f}.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('kernels_ms_per_step_untimed_pass'))
PY
