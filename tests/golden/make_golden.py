"""Generate tests/golden/calib_golden.json: golden input/output vectors for the calibration chain.

Inputs are regenerated from seeds by gsmcal.synth (the generator is part of the repo, so only the
seeds, a checksum of the raw bytes and the expected outputs are committed).  Expected outputs come
from oracle/gsmcal_oracle.py -- the reference itself is MATLAB and cannot run here (no MATLAB/Octave),
so these vectors pin the ORACLE and the HIP path against drift; they are not reference outputs
("parity unpinned")."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsmcal  # noqa: E402
from oracle import gsmcal_oracle as o  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refvec  # noqa: E402

synth = gsmcal.synth
FC = 957.4e6


def f(x):
    return "inf" if isinstance(x, float) and math.isinf(x) else x


def main():
    coef = o.fir1(46, 200e3 / synth.FS)
    coef30 = o.fir1(30, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    cases = []
    for dongle in range(8):
        raw, truth = synth.make_stream(dongle=dongle, arfcn=0, num_frames=102)
        out = o.calibrate_stream(raw, coef, ts, FC)
        cases.append({
            "dongle": dongle, "arfcn": 0, "num_frames": 102, "raw_sum": int(np.sum(raw.astype(np.uint64))),
            "truth_sampling_ppm": truth["sampling_ppm"], "truth_carrier_ppm": truth["carrier_ppm"],
            "coarse_pos": out["coarse_pos"].tolist(), "coarse_snr": out["coarse_snr"].tolist(),
            "fine_first_round_pos": out["fine_first_round_pos"].tolist(), "fcch_pos": out["fcch_pos"].tolist(),
            "sch_first_round_pos": out["sch_first_round_pos"].tolist(), "pos_info": out["pos_info"].ravel().tolist(),
            "sampling_ppm": [f(float(v)) for v in out["sampling_ppm"]],
            "carrier_ppm": [f(float(v)) for v in out["carrier_ppm"]],
            "total_sampling_ppm": f(out["total_sampling_ppm"]), "total_carrier_ppm": f(out["total_carrier_ppm"]),
            "r_len": out["r_len"],
            # SURVEY 8c front-end vectors: DC-removed checksum + first/last 16 values (raw2iq.m:5-8), FIR output at 32 probe
            # indices (gsm_sync_demod.m:110); 1-based indices, same record as tests/golden/make_reference_vectors.m writes
            "front_end": refvec.front_end_record(o, raw, coef)[0],
        })
    scans = []
    for arfcn in range(6):
        raw, truth = synth.make_stream(dongle=50, arfcn=arfcn, num_frames=64, bcch=(arfcn % 2 == 0))
        sc = o.scan_capture(raw, coef30)
        scans.append({"dongle": 50, "arfcn": arfcn, "num_frames": 64, "bcch": arfcn % 2 == 0,
                      "raw_sum": int(np.sum(raw.astype(np.uint64))), "coarse_pos": sc["coarse_pos"].tolist(),
                      "coarse_snr": sc["coarse_snr"].tolist(), "snr": sc["snr"], "num_hit": sc["num_hit"],
                      "front_end": refvec.front_end_record(o, raw, coef30)[0]})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "calib_golden.json"), "w") as fh:
        json.dump({"carrier_freq": FC, "seed": synth.DEFAULT_SEED, "generator": "tests/golden/make_golden.py",
                   "cases": cases, "scans": scans}, fh, indent=1)


if __name__ == "__main__":
    main()
