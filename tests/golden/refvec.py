"""Pinning the oracle against the reference ITSELF, for whoever has MATLAB (or Octave + signal/communications packages).

The build image has neither, so parity is "unpinned" (DESIGN.md 0).  This module makes pinning a one-command job:

    python tests/golden/refvec.py export  [DIR]      # seeded captures (uint8 .bin), manifest.txt, exported fir1 taps and SCH template
    matlab -batch "cd tests/golden; make_reference_vectors('/path/to/multi-rtl-sdr-calibration', 'captures')"
                                                     # -> tests/golden/reference_vectors.json (commit it)
    python -m pytest tests/test_reference_vectors_cpu.py     # both oracles against the reference's own outputs

`make_reference_vectors.m` is this repository's own script: it addpath()s a checkout of the reference (it contains no
reference code), feeds the exported captures through the reference's functions in the order of gsm_sync_demod.m:107-124 and
multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,164-185, and writes per-stage outputs.  `stage_vectors()` below computes the same
record with one of this repository's oracles; `compare()` lines the two up.  Nothing here travels to the GPU box or is
imported by the product."""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FC = 957.4e6                      # gsm_sync_demod.m:14
OV, DEC = 8, 8
# (name, kind, dongle, arfcn, frames, make_stream kwargs): small enough for MATLAB's toeplitz fine search (55 MB per hit)
CAPTURES = [
    ("sync_d0", "sync", 0, 0, 102, {}),
    ("sync_d3", "sync", 3, 0, 102, {}),
    ("sync_d5_lowsnr", "sync", 5, 0, 102, {"snr_db": 9.0}),
    ("sync_d6_bigppm", "sync", 6, 0, 102, {"sampling_ppm": 180.0, "carrier_ppm": -35.0}),
    ("sync_d7_nobcch", "sync", 7, 0, 102, {"bcch": False}),
    ("scan_a0", "scan", 50, 0, 64, {}),
    ("scan_a1_nobcch", "scan", 50, 1, 64, {"bcch": False}),
    ("scan_a2", "scan", 50, 2, 64, {}),
]
N_PROBE = 32


def probe_indices(n, count=N_PROBE):
    """0-based probe positions spread over [0, n): the same arithmetic in make_reference_vectors.m (floor((k*(n-1))/(count-1)))."""
    return [(k * (n - 1)) // (count - 1) for k in range(count)]


def capture(spec):
    from gsmcal import synth
    name, kind, dongle, arfcn, frames, kw = spec
    return synth.make_stream(dongle=dongle, arfcn=arfcn, num_frames=frames, **kw)[0]


def export(outdir):
    from gsmcal import synth
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "manifest.txt"), "w") as mf:
        for spec in CAPTURES:
            raw = capture(spec)
            raw.tofile(os.path.join(outdir, spec[0] + ".bin"))
            mf.write(f"{spec[0]} {spec[1]} {len(raw) // 2}\n")
    # what the harness falls back to where a toolbox is missing (it records which source it used)
    synth.fir1(46, 200e3 / synth.FS).astype("<f8").tofile(os.path.join(outdir, "fir1_46.f64"))
    synth.fir1(30, 200e3 / synth.FS).astype("<f8").tofile(os.path.join(outdir, "fir1_30.f64"))
    ts = np.asarray(synth.sch_training_sequence(), dtype=np.complex128)
    np.stack([ts.real, ts.imag], axis=1).astype("<f8").tofile(os.path.join(outdir, "sch_training_sequence_8x.f64"))
    return outdir


# ---- the record, computed with one of this repository's oracles ---------------------------------------------------------
def _cplx_probe(v, idx):
    v = np.asarray(v)
    return {"idx": [int(i) + 1 for i in idx], "re": [float(v[i].real) for i in idx], "im": [float(v[i].imag) for i in idx]}


def _vec(x):
    return [float(v) for v in np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel()]


def front_end_record(o, raw, coef):
    """raw2iq.m:5-8 and the drivers' filter(coef,1,.) (gsm_sync_demod.m:110): checksums, first/last 16, 32 probes."""
    r = o.raw2iq(np.asarray(raw, dtype=np.float64))
    if r.ndim == 2:
        r = r[:, 0]
    filt = o.matlab_filter(coef, r) if hasattr(o, "matlab_filter") else o.filter_fir(coef, r)
    n = len(r)
    rec = {"raw2iq": {"n": n, "sum_re": float(np.sum(r.real)), "sum_im": float(np.sum(r.imag)),
                      "sum_abs2": float(np.sum(r.real ** 2 + r.imag ** 2)),
                      "first16": _cplx_probe(r, range(16)), "last16": _cplx_probe(r, range(n - 16, n))},
           "filter": _cplx_probe(filt, probe_indices(n))}
    return rec, filt


def scanner_accept(pos, snr):
    """multi_rtl_sdr_gsm_FCCH_scanner.m:168-185 (driver glue, stated here once more so that both oracles share it): at
    least three hits, every spacing within 50 of 12500 or, failing that, of 12500 + 1250."""
    pos, snr = np.asarray(pos, dtype=np.float64), np.asarray(snr, dtype=np.float64)
    if len(pos) >= 3:
        d = np.diff(pos)
        off = np.abs(d - 12500.0) > 50.0
        if not off.any() or not (np.abs(d[off] - 13750.0) > 50.0).any():
            acc = 0.0
            for v in snr:                                     # mean() = left-to-right sum / n
                acc += float(v)
            return acc / len(snr), float(len(pos))
    return 0.0, 0.0


def stage_vectors(o, raw, kind, coef, ts, fc=FC):
    rec, rf = front_end_record(o, raw, coef)
    dec = rf[0::OV * DEC]
    pos, snr = o.FCCH_coarse_position(dec, DEC)
    rec["coarse_pos"], rec["coarse_snr"] = _vec(pos), _vec(snr)
    if kind == "scan":
        rec["snr"], rec["num_hit"] = scanner_accept(np.atleast_1d(pos), np.atleast_1d(snr))
        return rec
    fpos, r1, sp1, cp1 = o.FCCH_fine_correction(rf, pos, OV, fc)[:4]
    rec["fcch_pos"], rec["sampling_ppm1"], rec["carrier_ppm1"] = _vec(fpos), float(sp1), float(cp1)
    rec["r1_len"] = len(r1) if isinstance(r1, np.ndarray) and np.ndim(r1) else -1
    if rec["r1_len"] > 0:
        rec["r1_probe"] = _cplx_probe(r1, probe_indices(rec["r1_len"]))
    pi_ret, r2, sp2 = o.SCH_corr_rate_correction(r1, fpos, ts, OV)[:3]
    pi = np.atleast_2d(np.asarray(pi_ret, dtype=np.float64))
    rec["pos_info_rows"] = int(pi.shape[0])
    rec["pos_info"] = _vec(pi.T)                       # column-major, as MATLAB's pos_info(:) gives it
    rec["sampling_ppm2"] = float(sp2)
    rec["r2_len"] = len(r2) if isinstance(r2, np.ndarray) and np.ndim(r2) else -1
    r3, cp2 = o.carrier_correct_post_SCH(r2, pi_ret, OV, fc)[:2]
    rec["carrier_ppm2"] = float(cp2)
    rec["r3_len"] = len(r3) if isinstance(r3, np.ndarray) and np.ndim(r3) else -1
    if rec["r3_len"] > 0:
        rec["r3_probe"] = _cplx_probe(r3, probe_indices(rec["r3_len"]))
    rec["total_sampling_ppm"] = float(o.total_ppm_calculation(np.array([sp1, sp2])))
    rec["total_carrier_ppm"] = float(o.total_ppm_calculation(np.array([cp1, cp2])))
    return rec


# ---- reading the harness's file and lining the two up -----------------------------------------------------------------
def _num(v):
    if isinstance(v, str):
        return {"inf": math.inf, "-inf": -math.inf, "nan": math.nan}[v.lower()]
    return float(v)


def _nums(v):
    return np.array([_num(x) for x in (v if isinstance(v, list) else [v])], dtype=np.float64)


def load(path):
    with open(path) as f:
        return json.load(f)


def compare(ref, mine, what, ppm_rtol=1e-6, ppm_atol=1e-9, val_rtol=1e-9):
    """ref: one capture's record from the reference run; mine: stage_vectors().  Returns a list of mismatch strings.
    Integer positions must be identical; ppm within north_star's 1e-6 relative (1e-9 ppm floor); sample values and SNRs
    within val_rtol of the record's peak (MATLAB's filter / fft / interp1 round differently at the 1e-16 level)."""
    bad = []

    def close(a, b, rtol, atol, key):
        a, b = _nums(a), _nums(b)
        if a.shape != b.shape:
            bad.append(f"{what}.{key}: shape {b.shape} vs reference {a.shape}")
            return
        fin = np.isfinite(a)
        if not np.array_equal(fin, np.isfinite(b)) or not np.array_equal(a[~fin], b[~fin], equal_nan=True):
            bad.append(f"{what}.{key}: non-finite pattern differs ({b} vs reference {a})")
            return
        if fin.any() and not np.all(np.abs(a[fin] - b[fin]) <= rtol * np.abs(a[fin]) + atol):
            bad.append(f"{what}.{key}: {b[fin][:4]} vs reference {a[fin][:4]} (max abs diff {np.max(np.abs(a[fin] - b[fin])):.3e})")

    def exact(key):
        if key in ref:
            close(ref[key], mine.get(key, []), 0.0, 0.0, key)

    def probe(key):
        if key in ref and key in mine:
            scale = max(1e-300, float(np.max(np.abs(_nums(ref[key]["re"])))), float(np.max(np.abs(_nums(ref[key]["im"])))))
            if [int(i) for i in ref[key]["idx"]] != mine[key]["idx"]:
                bad.append(f"{what}.{key}: probe indices differ")
                return
            close(ref[key]["re"], mine[key]["re"], 0.0, val_rtol * scale, key + ".re")
            close(ref[key]["im"], mine[key]["im"], 0.0, val_rtol * scale, key + ".im")

    if "raw2iq" in ref:
        for k in ("sum_re", "sum_im", "sum_abs2"):
            close(ref["raw2iq"][k], mine["raw2iq"][k], 1e-9, 1e-3, "raw2iq." + k)
        for k in ("first16", "last16"):
            for c in ("re", "im"):
                close(ref["raw2iq"][k][c], mine["raw2iq"][k][c], 0.0, 1e-12, f"raw2iq.{k}.{c}")   # integers minus a mean: exact to rounding
    probe("filter")
    exact("coarse_pos")
    if "coarse_snr" in ref:
        close(ref["coarse_snr"], mine["coarse_snr"], 0.0, 1e-8, "coarse_snr")
    for k in ("fcch_pos", "pos_info", "pos_info_rows", "r1_len", "r2_len", "r3_len", "num_hit"):
        exact(k)
    if "snr" in ref:
        close(ref["snr"], mine["snr"], 0.0, 1e-8, "snr")
    for k in ("sampling_ppm1", "carrier_ppm1", "sampling_ppm2", "carrier_ppm2", "total_sampling_ppm", "total_carrier_ppm"):
        if k in ref:
            close(ref[k], mine[k], ppm_rtol, ppm_atol, k)
    probe("r1_probe")
    probe("r3_probe")
    return bad


def write_with_oracle(path, o, outdir=None):
    """The harness's file as THIS repository's oracle would write it (self-test of the comparison; never a pin)."""
    from gsmcal import synth
    coef46, coef30 = synth.fir1(46, 200e3 / synth.FS), synth.fir1(30, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    doc = {"generator": "tests/golden/refvec.py write_with_oracle (NOT the reference)", "captures": {}}
    for spec in CAPTURES:
        raw = capture(spec)
        doc["captures"][spec[0]] = stage_vectors(o, raw, spec[1], coef46 if spec[1] == "sync" else coef30, ts)
    with open(path, "w") as f:
        json.dump(_jsonable(doc), f)
    return doc


def _jsonable(v):
    """non-finite numbers as the strings the MATLAB harness writes ("inf", "-inf", "nan"): strict JSON has no Infinity"""
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, float) and not math.isfinite(v):
        return "nan" if math.isnan(v) else ("inf" if v > 0 else "-inf")
    return v


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "export":
        print(export(sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "captures")))
    else:
        print(__doc__)
