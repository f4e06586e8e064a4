"""Constructed worst cases under -m gpu: the certified shortcuts of the HIP path -- the fine search's 8-bin
certificate + fp32 chunk sweep + fp64 verify, the coarse detector's certified scan with its exact serial replay, the
DC removal by linearity -- against the oracle on inputs built so that the reference's first-max / strict-threshold rules
decide by a hair (tests/worstcase.py; tests/test_worstcase_cpu.py proves the constructions in the oracle's arithmetic).
Everything goes through the C ABI; GSMCAL_CERT=0 (every chunk swept) and GSMCAL_PRESCREEN=0 (plain all-bin fp64 search)
must give the same answers."""
import math

import numpy as np
import pytest

import parity
import worstcase as wc
from oracle import gsmcal_oracle as o

pytestmark = pytest.mark.gpu
FC = 957.4e6


@pytest.fixture(scope="module")
def g(gsmcal_mod, ctx):
    return gsmcal_mod


@pytest.fixture(scope="module")
def fine_contexts(g):
    """default path, no certificate, no prescreen: contexts read the environment when they are created"""
    import os
    made = {"default": g.default_context(0)}
    for name, var in (("no_cert", "GSMCAL_CERT"), ("no_prescreen", "GSMCAL_PRESCREEN")):
        os.environ[var] = "0"
        try:
            made[name] = g.Context(0)
        finally:
            del os.environ[var]
    yield made
    for name in ("no_cert", "no_prescreen"):
        made[name].close()


# ---- (a), (b), (e): two-tone near-ties in the fine search ------------------------------------------------------------
@pytest.mark.parametrize("family", ["interior", "edge"])
@pytest.mark.parametrize("eps", [1e-6, -1e-6, 1e-9, -1e-9, 1e-12, -1e-12])
def test_fine_search_two_tone_near_ties(g, fine_contexts, family, eps):
    """FCCH_fine_correction.m:48-52: argmax over 1025 shifts x 1184 bins with two peaks a factor 1 + eps apart -- in bins the
    certificate does not evaluate, 64 to 1024 shifts apart, in interior and in edge chunks, the larger one first or second."""
    geoms = wc.TWO_TONE_GEOMETRIES[family]
    s, base, want = wc.two_tone_stream([gm + (eps,) for gm in geoms])
    info = {}
    o_fp, _, o_sp, o_cp = o.FCCH_fine_correction(s, base, 8, FC, info=info)
    assert np.array_equal(info["first_round_pos"], want)              # the construction holds in the oracle
    for name, cx in fine_contexts.items():
        fp, _, sp, cp = g.FCCH_fine_correction(s, base, 8, FC, ctx=cx, want_r=False)
        det = g.last_batch_details(1, ctx=cx)
        n = det["counts"][0][1]
        parity.assert_positions(det["fine_first"][0, :n], want, f"first-round FCCH_pos ({name}, eps {eps:g})")
        parity.assert_positions(fp, o_fp, f"FCCH_pos ({name})")
        parity.assert_ppm(sp, o_sp, f"sampling ppm ({name})")
        parity.assert_ppm(cp, o_cp, f"carrier ppm ({name})")


# ---- (c): coarse decisions within 1e-7 .. 1e-10 dB of the threshold ---------------------------------------------------
@pytest.mark.parametrize("where", ["first", "hop"])
@pytest.mark.parametrize("delta", [1e-5, -1e-5, 2e-6, -2e-6, 1e-7, -1e-7, 1e-9, -1e-9, 1e-10, -1e-10])
def test_coarse_decisions_on_the_threshold(g, where, delta):
    """move_fft_snr_runtime_avg.m:30-32 / specific_fft_snr_fix_avg.m:24-26: snr - avg - th = +-delta at the first candidate
    window / at a hop candidate.  |delta| <= 1e-6 is inside the certified scan's margin (the stream must take the exact
    serial replay), 2e-6 and 1e-5 just outside it (decided by the certificate).  An exact zero cannot be built: the SNR values
    of any two implementations differ by ~1e-13 dB, +-1e-10 brackets it with room for that."""
    s, info = wc.coarse_threshold_stream(delta, where)
    want_p, want_s = o.FCCH_coarse_position(s, 8)
    got_p, got_s = g.FCCH_coarse_position(s, 8)
    parity.assert_positions(got_p, want_p, f"coarse positions ({where}, margin {info['margin']:g})")
    assert np.allclose(got_s, want_s, rtol=0, atol=parity.SNR_ATOL)
    if where == "first":
        want = o.move_fft_snr_runtime_avg(s[:3594], 160, 16, 10)
        got = g.move_fft_snr_runtime_avg(s[:3594], 160, 16, 10)
        assert got[:2] == want[:2] and abs(got[2] - want[2]) < parity.SNR_ATOL and abs(got[3] - want[3]) < parity.SNR_ATOL


# ---- (d): degenerate captures through both batch paths ----------------------------------------------------------------
@pytest.fixture(scope="module")
def setup(g):
    s = g.synth
    return {"coef": s.fir1(46, 200e3 / s.FS), "coef30": s.fir1(30, 200e3 / s.FS), "ts": s.sch_training_sequence()}


def test_degenerate_captures_match_the_oracle(g, setup):
    """Constant bytes (every window SNR is 0/0 = NaN in the reference: the DC removal by linearity must not turn the
    rounding residue into a spectrum), both rails, rail-to-rail square wave, hard-clipped noise, a full-scale CW that never
    turns off: calibration chain and scanner path against the oracle, all in one batch with a normal stream between them."""
    caps = wc.degenerate_captures(102)
    names = list(caps)
    normal = g.synth.make_stream(dongle=3)[0]
    raw = np.stack([caps[k] for k in names[:3]] + [normal] + [caps[k] for k in names[3:]])
    out = g.calibrate_batch(raw, setup["coef"], setup["ts"], FC)
    det = g.last_batch_details(len(raw))
    for i in range(len(raw)):
        with np.errstate(all="ignore"):
            orc = o.calibrate_stream(raw[i], setup["coef"], setup["ts"], FC)
        parity.compare_stream(orc, out["table"][i], det, i, out["pos_info"][i])
    sc = g.fcch_scan_batch(np.ascontiguousarray(raw[:, : 2 * 640000]), setup["coef30"])
    for i in range(len(raw)):
        with np.errstate(all="ignore"):
            live = o.scan_capture(raw[i, : 2 * 640000], setup["coef30"])
        assert live["num_hit"] == sc["num_hit"][i] and abs(live["snr"] - sc["snr"][i]) < parity.SNR_ATOL, names
        n = sc["counts"][i]
        if np.ndim(live["coarse_pos"]) and live["coarse_pos"][0] != -1.0:
            parity.assert_positions(sc["positions"][i, :n], live["coarse_pos"], "scan positions")
        else:
            assert n == 0


def test_constant_stretches_inside_a_live_capture(g, setup):
    """Dropped samples filled with a constant, before the first FCCH and inside the moving search's range.  Filled with the
    DC level rounded to a byte the windows inside have exactly zero noise (SNR +inf: the reference hits there); in a capture
    whose mean is exactly that byte they are exact zeros (SNR 0/0 = NaN, which then sits in the reference's running sum so
    that no later window can hit: move_fft_snr_runtime_avg.m:30-41).  The batch path removes the DC term by linearity
    (rounding residue ~1e-14 instead of exact zeros) and must still land on the reference's answer in every case."""
    caps = wc.constant_stretch_captures(g.synth.make_stream(dongle=3)[0])
    names = list(caps)
    batch = np.stack([caps[k] for k in names])
    out = g.calibrate_batch(batch, setup["coef"], setup["ts"], FC)
    det = g.last_batch_details(len(batch))
    kinds = set()
    for i in range(len(batch)):
        with np.errstate(all="ignore"):
            orc = o.calibrate_stream(batch[i], setup["coef"], setup["ts"], FC)
            r = o.matlab_filter(setup["coef"], o.raw2iq(batch[i].astype(np.float64)))[0::64]
            sn = wc.snr_series(r[:3594])
        kinds.add("nan" if np.any(np.isnan(sn)) else ("inf" if np.any(np.isinf(sn)) else "finite"))
        parity.compare_stream(orc, out["table"][i], det, i, out["pos_info"][i])
    assert {"nan", "inf"} <= kinds, f"the constructions must produce both degenerate SNR values in the oracle: {kinds}"


def test_half_bin_fcch_tones_follow_the_oracle(g, setup):
    """FCCH tone exactly half-way between two bins of the 1184-point transform (carrier offset = (k + 1/2) fs/1184): the
    flat-topped peak is where the reference's fine search is ill-conditioned; whatever it does -- positions or its
    sentinels -- the HIP path must do the same."""
    fs = g.synth.FS
    raws = []
    for d, k in enumerate((0, 1, -2, 3, -5, 7)):
        ppm = (k + 0.5) * (fs / 1184.0) / FC * 1e6
        raws.append(g.synth.make_stream(dongle=200 + d, carrier_ppm=ppm)[0])
    raw = np.stack(raws)
    out = g.calibrate_batch(raw, setup["coef"], setup["ts"], FC)
    det = g.last_batch_details(len(raw))
    for i in range(len(raw)):
        orc = o.calibrate_stream(raw[i], setup["coef"], setup["ts"], FC)
        parity.compare_stream(orc, out["table"][i], det, i, out["pos_info"][i])


# ---- graphs of two batch geometries alternating on one context ------------------------------------------------
def test_forced_graphs_with_alternating_capture_lengths(g, setup, monkeypatch):
    """GSMCAL_GRAPH=2 captures the single-lane fused chain.  Two capture lengths (102 and 61 frames: 12 and 8 windows per
    stream) alternate on one context; the exchange block of k_post_chain_r must serve both layouts, eager and replayed."""
    a = np.stack([g.synth.make_stream(dongle=d, num_frames=102)[0] for d in (0, 3)])
    b = np.stack([g.synth.make_stream(dongle=d, num_frames=61)[0] for d in (20, 21, 22)])
    ref_a = g.calibrate_batch(a, setup["coef"], setup["ts"], FC)["table"]
    ref_b = g.calibrate_batch(b, setup["coef"], setup["ts"], FC)["table"]
    monkeypatch.setenv("GSMCAL_GRAPH", "2")
    cx = g.Context(0)
    monkeypatch.delenv("GSMCAL_GRAPH")
    try:
        order = "aaabbbababaabbab"                          # eager, capture, replay of each, then interleaved replays
        for ch in order:
            raw, want = (a, ref_a) if ch == "a" else (b, ref_b)
            out = g.calibrate_batch(raw, setup["coef"], setup["ts"], FC, ctx=cx)
            assert np.array_equal(out["table"], want, equal_nan=True), f"call '{ch}' of {order}"
    finally:
        cx.close()
