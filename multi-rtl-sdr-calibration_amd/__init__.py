"""gsmcal -- MI355X-native GSM FCCH+SCH calibration chain (package dir: multi-rtl-sdr-calibration_amd/).

Import it as `import gsmcal` (the top-level gsmcal.py aliases this directory, whose name is not a
valid Python identifier).  Sub-modules:
    api    -- host mirror of the reference's MATLAB functions + batched entry points (ctypes -> HIP)
    synth  -- seeded synthetic GSM IQ, GMSK modulator, fir1
    dist   -- multi-GPU sharding + all-gather of the calibration table (torch.distributed, RCCL)
    ingest -- rtl_tcp framing (command packets, flush, capture reads) + the pinned ring / async H2D of the C ABI
    build  -- hipcc build of csrc/ -> lib/libgsmcal.so
"""
from . import build, ingest, synth  # noqa: F401
from ._lib import GsmcalError, MAX_HITS, MAX_POS_ROWS, TABLE_COLS, SIGNATURES, lib_path, load  # noqa: F401
from .api import (  # noqa: F401
    Context, default_context, raw2iq, filter, chn_filter_8x_4x, chn_filter_4x, move_fft_snr_runtime_avg,
    specific_fft_snr_fix_avg, FCCH_coarse_position, FCCH_fine_correction, SCH_corr_rate_correction,
    carrier_correct_post_SCH, total_ppm_calculation, SCH_equalise, frontend_batch, fcch_scan_batch, calibrate_batch,
    last_batch_details, last_batch_snr, TABLE_FIELDS, fcch_scan_batch_dev, calibrate_batch_dev, synth_expand_dev,
    set_verbose, last_call_report, num2str,
)
