mkdir -p gpurun_out
bash tools/abn.sh multi-rtl-sdr-calibration_amd/lib/exp_prev.so multi-rtl-sdr-calibration_amd/lib/libgsmcal.so
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
