"""Largest difference between the SNR table of the batch path and the oracle's window SNRs (dB), per stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gsmcal as g
from oracle import gsmcal_oracle as o
coef = g.synth.fir1(46, 200e3 / g.synth.FS); ts = g.synth.sch_training_sequence()
kws = [{}, {}, {}, {}, {"snr_db": 7.0}, {"bcch": False}, {"snr_db": 28.0}, {"sampling_ppm": 250.0}]
raw = np.stack([g.synth.make_stream(dongle=40 + i, **kw)[0] for i, kw in enumerate(kws)])
g.calibrate_batch(raw, coef, ts, 957.4e6)
for i in range(len(raw)):
    tab, n_mov = g.last_batch_snr(i)
    s = o.matlab_filter(coef, o.raw2iq(raw[i].astype(np.float64)))[0::64]
    want = o._window_snr(o._power_spectra(s, 1, len(s) - 15, 16))
    ok = np.isfinite(tab[:len(want)])
    d = np.abs(tab[:len(want)][ok] - want[ok])
    print(f"stream {i} {kws[i]}: windows {len(want)}, computed {ok.sum()}, max |dSNR| {d.max():.3e} dB, median {np.median(d):.1e}")
