"""CPU tests of the oracle (oracle/gsmcal_oracle.py): MATLAB semantics, the data the reference pins
(filter taps, constants), known-answer behaviour on synthetic GSM, and the committed golden vectors."""
import json
import math
import os
import sys

import numpy as np
import pytest

from oracle import gsmcal_oracle as o

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def synth():
    import gsmcal
    return gsmcal.synth


def test_matlab_round_half_away_from_zero():
    # FCCH_coarse_position.m:35-36: round(1562.5) = 1563, round(1718.75) = 1719
    assert o.matlab_round(12500 / 8) == 1563 and o.matlab_round(13750 / 8) == 1719
    assert o.matlab_round(-2.5) == -3 and o.matlab_round(2.5) == 3 and o.matlab_round(0.49999) == 0


def test_channel_filter_taps_pinned_by_fda():
    # refnum of gsm_chn_filter_8x.fda -- 60 taps, exactly symmetric, first/centre values
    num = o.load_num(os.path.join(GOLD, "gsm_chn_filter_8x_num.txt"))
    assert num.shape == (60,) and np.array_equal(num, num[::-1])
    assert num[0] == -8.3045994016978379e-04 and num[29] == num[30] == 1.3251507266392798e-01
    assert abs(num.sum() - 0.99423319989488024) < 1e-15
    num4 = o.load_num(os.path.join(GOLD, "gsm_chn_filter_4x_num.txt"))
    assert num4.shape == (30,) and num4[0] == -2.1523104568354720e-03 and num4[14] == 2.5886862810293920e-01


def test_fir1_is_hamming_lowpass_unit_dc(synth):
    for n in (46, 30):
        h = o.fir1(n, 200e3 / synth.FS)
        assert len(h) == n + 1 and abs(h.sum() - 1) < 1e-14 and np.allclose(h, h[::-1], atol=1e-17)
        assert np.max(np.abs(h - synth.fir1(n, 200e3 / synth.FS))) < 1e-16  # independent statement agrees


def test_raw2iq_removes_exact_mean():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(2000, 3)).astype(np.float64)
    b = o.raw2iq(a)
    c = a[0::2] + 1j * a[1::2]
    assert b.shape == (1000, 3)
    tot = c.sum(axis=0)
    assert np.array_equal(b, c - (tot.real / 1000 + 1j * (tot.imag / 1000)))   # complex ./ real, part by part
    assert np.max(np.abs(b.mean(axis=0))) < 1e-12


def test_filter_is_causal_zero_state_and_decimator_keeps_odd_rows():
    x = np.zeros(100, dtype=complex)
    x[0] = 1
    num = o.load_num(os.path.join(GOLD, "gsm_chn_filter_8x_num.txt"))
    y = o.matlab_filter(num, x)
    assert np.allclose(y[:60].real, num) and np.all(y[60:] == 0)          # impulse response, no delay comp.
    r = o.chn_filter_8x_4x(x, num)
    assert len(r) == 50 and np.allclose(r.real[:30], num[0::2])            # r(1:2:end)

def test_chn_filter_4x_is_a_plain_causal_filter():
    # chn_filter_4x.m:13: r = filter(coef,1,s), all rows kept, zero initial state
    num4 = o.load_num(os.path.join(GOLD, "gsm_chn_filter_4x_num.txt"))
    x = np.zeros(64, dtype=np.complex128)
    x[3] = 1.0 + 2.0j
    r = o.chn_filter_4x(x, num4)
    assert r.shape == x.shape and np.all(r[:3] == 0)
    assert np.allclose(r[3:33], (1.0 + 2.0j) * num4, rtol=0, atol=1e-18)


def test_toeplitz_slice_equals_sliding_windows():
    # FCCH_fine_correction.m:48-49: toeplitz(...)(len:end, end:-1:1) column k == s(sp+k-1 : sp+k-1+fft_len-1)
    from scipy.linalg import toeplitz
    rng = np.random.default_rng(2)
    length, fft_len = 9, 6
    seg = rng.standard_normal(length + fft_len - 1) + 1j * rng.standard_normal(length + fft_len - 1)
    m = toeplitz(seg, np.concatenate([[seg[0]], np.zeros(length - 1)]))
    m = m[length - 1:, ::-1]
    win = np.lib.stride_tricks.sliding_window_view(seg, fft_len)
    assert np.array_equal(m, win.T)


def test_move_fft_incremental_average_and_first_hit():
    rng = np.random.default_rng(3)
    n = 1200
    s = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    s[700:760] += 8 * np.exp(2j * np.pi * 0.11 * np.arange(60))
    hf, hi, havg, hs = o.move_fft_snr_runtime_avg(s, 160, 16, 10)
    assert hf and 680 <= hi <= 705
    assert hs - havg > 10
    # no hit possible inside the first mv_len windows: the average is seeded with 999 (:11)
    s2 = s.copy()
    s2[20:80] += 8 * np.exp(2j * np.pi * 0.11 * np.arange(60))
    hf2, hi2, _, _ = o.move_fft_snr_runtime_avg(s2, 160, 16, 10)
    assert hf2 and hi2 > 160
    # sentinel
    assert o.move_fft_snr_runtime_avg(rng.standard_normal(400) + 0j, 160, 16, 10) == (False, -1, math.inf, math.inf)


def test_specific_fft_window_bounds_raise_like_matlab():
    from oracle import gsmcal_oracle_literal as lit
    rng = np.random.default_rng(5)
    s = rng.standard_normal(100) + 1j * rng.standard_normal(100)
    with pytest.raises(o.MatlabIndexError):
        o.specific_fft_snr_fix_avg(s, (0, 5), 16, 10, 0.0)
    with pytest.raises(o.MatlabIndexError):          # windows 80..85 fit and miss (noise), 86 runs past the end
        o.specific_fft_snr_fix_avg(s, (80, 90), 16, 10, 50.0)
    with pytest.raises(IndexError):
        lit.specific_fft_snr_fix_avg(s, (80, 90), 16, 10, 50.0)
    # the reference indexes window by window (specific_fft_snr_fix_avg.m:10-11): a hit in a window that fits is returned before
    # the loop reaches a window that would run past the end -- both restatements agree
    s2 = s.copy()
    s2[82:98] += 40 * np.exp(2j * np.pi * 0.125 * np.arange(16))
    a = o.specific_fft_snr_fix_avg(s2, (80, 90), 16, 10, 0.0)
    b = lit.specific_fft_snr_fix_avg(s2, (80, 90), 16, 10, 0.0)
    assert a[0] and b[0] and a[1] == b[1] and 80 <= a[1] <= 85 and abs(a[2] - b[2]) < 1e-9
    # lo itself out of range: the very first iteration is the index error
    with pytest.raises(o.MatlabIndexError):
        o.specific_fft_snr_fix_avg(s2, (86, 90), 16, 10, 0.0)


def test_total_ppm_calculation():
    assert o.total_ppm_calculation([np.inf, np.inf]) == math.inf
    assert o.total_ppm_calculation([10.0, np.inf]) == math.inf
    v = o.total_ppm_calculation([34.78260869565217, -1.0869565217391304])
    assert abs(v - ((1 + 34.78260869565217e-6) * (1 - 1.0869565217391304e-6) - 1) * 1e6) < 1e-12


def test_sentinels_propagate_like_the_reference():
    # <5 hits -> FCCH_pos=-1, r=-1, ppm=inf (FCCH_fine_correction.m:8-15)
    fp, r, sp, cp = o.FCCH_fine_correction(np.zeros(200000, complex), [100, 200, 300], 8, 957.4e6)
    assert fp == -1.0 and r == -1.0 and sp == math.inf and cp == math.inf
    pi, r, sp = o.SCH_corr_rate_correction(-1.0, -1.0, np.ones(512, complex), 8)
    assert np.all(pi == -1) and r == -1.0 and sp == math.inf
    r, cp = o.carrier_correct_post_SCH(-1.0, pi, 8, 957.4e6)
    assert r == -1.0 and cp == math.inf
    # pos_info without 4 BCCH rows (carrier_correct_post_SCH.m:15-19)
    r, cp = o.carrier_correct_post_SCH(np.zeros(10, complex), np.array([[1, 0], [100, 1], [200, 2]]), 8, 957.4e6)
    assert r == -1.0 and cp == math.inf


def test_scanner_acceptance_rule():
    # multi_rtl_sdr_gsm_FCCH_scanner.m:168-185
    assert o.scanner_accept([1, 12501, 25001], [10.0, 12.0, 14.0]) == (12.0, 3.0)
    assert o.scanner_accept([1, 12501, 26251], [10.0, 12.0, 14.0]) == (12.0, 3.0)     # across the idle frame
    assert o.scanner_accept([1, 12501, 25301], [10.0, 12.0, 14.0]) == (0.0, 0.0)      # 300 off
    assert o.scanner_accept([1, 12501], [10.0, 12.0]) == (0.0, 0.0)                   # <3 hits
    assert o.scanner_accept(-1.0, -1.0) == (0.0, 0.0)


@pytest.mark.parametrize("sppm,cppm", [(25.0, -10.0), (-60.0, 18.5)])
def test_known_answer_recovers_injected_ppm(synth, sppm, cppm):
    """Known-answer test replacing the reference's missing tests: injected sampling and
    carrier errors come back within the position quantisation (~1.1 ppm) / estimator bias (<1 ppm)."""
    fc = 957.4e6
    # choose the carrier error so the FCCH tone sits on an FFT bin (the reference's fine search is only
    # well-conditioned there): tone = 67708.33 + cppm*957.4 Hz; bin width 1830 Hz
    raw, truth = synth.make_stream(dongle=100, sampling_ppm=sppm, carrier_ppm=cppm, snr_db=25.0, start_frame=3,
                                   frac_start=1234.5)
    coef = o.fir1(46, 200e3 / synth.FS)
    out = o.calibrate_stream(raw, coef, synth.sch_training_sequence(), fc)
    assert not math.isinf(out["total_sampling_ppm"]), "the oracle rejects a clean, on-bin, 25 dB stream: something regressed"
    assert abs(out["total_sampling_ppm"] - sppm) < 1.5
    assert abs(out["total_carrier_ppm"] - cppm) < 1.5
    d = np.diff(out["fcch_pos"])
    assert set(d.tolist()) <= {100000.0, 110000.0}
    pi = out["pos_info"]
    # burst map: FCCH then SCH one frame later; BCCH rows only right after a multiframe start
    f = pi[pi[:, 1] == 0, 0]
    s = pi[pi[:, 1] == 1, 0]
    assert np.all(s - f[:len(s)] == 10000)


def test_golden_vectors_reproduce():
    """Committed golden outputs (tests/golden/calib_golden.json, made by make_golden.py from seeded
    synthetic input) still come out of the oracle bit-for-bit on positions and to 1e-12 on ppm."""
    import gsmcal
    synth = gsmcal.synth
    with open(os.path.join(GOLD, "calib_golden.json")) as f:
        gold = json.load(f)
    coef = o.fir1(46, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    for case in gold["cases"][:2]:
        raw, _ = synth.make_stream(dongle=case["dongle"], arfcn=case["arfcn"], num_frames=case["num_frames"])
        assert int(np.sum(raw.astype(np.uint64))) == case["raw_sum"], "synthetic generator drifted"
        out = o.calibrate_stream(raw, coef, ts, gold["carrier_freq"])
        assert np.array_equal(out["coarse_pos"], np.asarray(case["coarse_pos"]))
        assert np.array_equal(out["fine_first_round_pos"], np.asarray(case["fine_first_round_pos"]))
        assert np.array_equal(out["pos_info"], np.asarray(case["pos_info"]).reshape(-1, 2))
        for k in ("total_sampling_ppm", "total_carrier_ppm"):
            g = case[k]
            if g == "inf":
                assert math.isinf(out[k])
            else:
                assert abs(out[k] - g) <= 1e-9 * max(1.0, abs(g))
        # SURVEY 8c front-end vectors: DC-removed checksums, first / last 16 values, FIR output at 32 probe indices
        sys.path.insert(0, GOLD)
        import refvec
        fe = refvec.front_end_record(o, raw, coef)[0]
        assert refvec.compare(case["front_end"], fe, f"dongle {case['dongle']}", val_rtol=1e-12) == []
    coef30 = o.fir1(30, 200e3 / synth.FS)
    sc = gold["scans"][0]
    raw, _ = synth.make_stream(dongle=sc["dongle"], arfcn=sc["arfcn"], num_frames=sc["num_frames"], bcch=sc["bcch"])
    assert refvec.compare(sc["front_end"], refvec.front_end_record(o, raw, coef30)[0], "scan 0", val_rtol=1e-12) == []


# ---- second, literal restatement (oracle/gsmcal_oracle_literal.py) against the vectorised oracle ------------------------
def _literal_chain(lit, raw, coef, ts, fc, ov=8, dec=8):
    r = lit.filter_fir(coef, lit.raw2iq(raw.astype(np.float64))[:, 0])
    pos, snr = lit.FCCH_coarse_position(r[0::ov * dec], dec)
    fp, r1, sp1, cp1, first = lit.FCCH_fine_correction(r, pos, ov, fc)
    pi, r2, sp2 = lit.SCH_corr_rate_correction(r1, fp, ts, ov)
    r3, cp2 = lit.carrier_correct_post_SCH(r2, pi, ov, fc)
    return {"coarse_pos": np.atleast_1d(pos), "coarse_snr": np.atleast_1d(snr), "first": np.asarray(first, dtype=np.float64),
            "fcch_pos": np.atleast_1d(np.asarray(fp, dtype=np.float64)), "pos_info": pi, "sp": [sp1, sp2], "cp": [cp1, cp2],
            "tot": [lit.total_ppm_calculation([sp1, sp2]), lit.total_ppm_calculation([cp1, cp2])], "r": r3}


@pytest.mark.parametrize("dongle", [0, 3, 2])
def test_literal_restatement_agrees_with_the_vectorised_oracle(synth, dongle):
    """Two independent restatements of the nine .m files -- vectorised (sliding windows, batched FFT, np.interp) and
    literal (scalar loops, explicit toeplitz + slice, definition DFTs, array-shift moving average, written-out interp1)
    -- give the same positions bit for bit and the same ppm to 1e-9 relative on full-size streams, including one the
    algorithm rejects (dongle 2).  This guards the oracle against a slip in ONE restatement; it does not pin MATLAB."""
    from oracle import gsmcal_oracle_literal as lit
    fc = 957.4e6
    raw, _ = synth.make_stream(dongle=dongle)
    coef = o.fir1(46, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    a = o.calibrate_stream(raw, coef, ts, fc, keep_r=True)
    b = _literal_chain(lit, raw, coef, ts, fc)
    assert np.array_equal(a["coarse_pos"], b["coarse_pos"])
    assert np.allclose(a["coarse_snr"], b["coarse_snr"], rtol=0, atol=1e-9)
    assert np.array_equal(a["fine_first_round_pos"], b["first"])
    assert np.array_equal(a["fcch_pos"], b["fcch_pos"])
    assert a["pos_info"].shape == b["pos_info"].shape and np.array_equal(a["pos_info"], b["pos_info"])
    for x, y in zip(list(a["sampling_ppm"]) + list(a["carrier_ppm"]) + [a["total_sampling_ppm"], a["total_carrier_ppm"]],
                    b["sp"] + b["cp"] + b["tot"]):
        assert (math.isinf(x) and math.isinf(y)) or abs(x - y) <= 1e-9 * abs(x) + 1e-12, (x, y)
    if isinstance(b["r"], np.ndarray):
        assert len(a["r_correct"]) == len(b["r"])
        assert np.max(np.abs(a["r_correct"] - b["r"])) <= 2e-9 * np.max(np.abs(b["r"]))
    else:
        assert a["r_len"] == -1


def test_literal_building_blocks_on_small_inputs():
    """move_fft / specific_fft (definition DFTs, shifting history), the toeplitz slice and interp1 on small random inputs."""
    from oracle import gsmcal_oracle_literal as lit
    rng = np.random.default_rng(3)
    s = rng.standard_normal(900) + 1j * rng.standard_normal(900)
    s[500:520] += 6 * np.exp(1j * 0.7 * np.arange(20))             # a tone burst the detector should hit
    for mv, L, th in ((40, 16, 8.0), (24, 8, 5.0), (50, 16, 99.0)):
        a, b = o.move_fft_snr_runtime_avg(s, mv, L, th), lit.move_fft_snr_runtime_avg(s, mv, L, th)
        assert a[0] == b[0] and a[1] == b[1]
        assert (math.isinf(a[2]) and math.isinf(b[2])) or (abs(a[2] - b[2]) < 1e-9 and abs(a[3] - b[3]) < 1e-9)
    a, b = o.specific_fft_snr_fix_avg(s, (480, 520), 16, 8.0, 0.0), lit.specific_fft_snr_fix_avg(s, (480, 520), 16, 8.0, 0.0)
    assert a[:2] == b[:2] and abs(a[2] - b[2]) < 1e-9
    v = rng.standard_normal(50) + 1j * rng.standard_normal(50)
    xq = np.arange(45) * 1.00037
    assert np.max(np.abs(lit.interp1_linear_unit_grid(v, xq) - o._interp1_linear(v, xq))) < 1e-14
    assert lit.m_round(1562.5) == 1563 and lit.m_round(-2.5) == -3 and o.matlab_round(1562.5) == 1563.0


def test_fine_search_rejection_rate_on_bin_vs_half_bin(synth):
    """DESIGN.md section 6 finding (1): the reference's fine search (FCCH_fine_correction.m:48-52) looks for the window with
    the largest single-bin power.  With the FCCH tone ON a bin of the 1184-point grid the peak over window starts is sharp;
    half-way BETWEEN two bins the energy splits, the peak flattens and noise picks the argmax, the first-round spacings
    leave the +-400-sample classes (:88-102) and the stream is rejected.  Measured here with the oracle alone: same
    streams, carrier offset placed on a bin or half a bin off."""
    fc = 957.4e6
    coef = o.fir1(46, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    binw = synth.FS / 1184.0                                         # 1830 Hz
    tone = synth.SYMBOL_RATE / 4.0                                   # 67 708.33 Hz = bin 37.0 exactly
    rej = {"on": 0, "half": 0}
    n = 6
    for i in range(n):
        for name, frac in (("on", 0.0), ("half", 0.5)):
            f_off = (i - n // 2) * binw + frac * binw                # whole bins away from 37, or half a bin further
            cppm = f_off / fc * 1e6
            raw, _ = synth.make_stream(dongle=300 + i, carrier_ppm=cppm, sampling_ppm=10.0 * (i - 3), snr_db=22.0)
            out = o.calibrate_stream(raw, coef, ts, fc)
            rej[name] += math.isinf(out["total_sampling_ppm"])
    assert rej["on"] == 0, f"on-bin streams must calibrate, {rej}"
    assert rej["half"] >= n // 2, f"half-bin streams are expected to be (mostly) rejected by the reference algorithm, {rej}"
