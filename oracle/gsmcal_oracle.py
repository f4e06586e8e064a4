"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GSM calibration DSP chain.

fp64 NumPy/SciPy restatement of the reference's nine hot-path MATLAB functions plus the driver
glue that calls them.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module; the product path (multi-rtl-sdr-calibration_amd/) never does.

PARITY UNPINNED: the reference is MATLAB source with no tests, golden vectors or captures, and
neither MATLAB nor Octave exists in the build container, so this restatement could not be checked
against reference outputs.  It follows the .m files statement by statement (citations below are
relative to the reference repo root), with MATLAB semantics made explicit:
  * round() is half-away-from-zero                       -> matlab_round
  * max() returns the FIRST maximal index                -> np.argmax (first occurrence)
  * positions are 1-based doubles                        -> kept 1-based in every interface
  * filter(b,1,x) is causal, zero initial state          -> scipy.signal.lfilter
  * interp1(...,'linear') on complex = lerp of re and im -> np.interp on .real/.imag
  * x' is the conjugate transpose
  * `if v == c` on an array is true only if ALL elements match
Third-party arithmetic that is not in the reference repo (MathWorks built-ins, version unpinned):
filter, fft, interp1, toeplitz, max, angle, log10, fir1 (Signal Processing Toolbox).  Open
equivalents used here: scipy.signal.lfilter, numpy.fft.fft (pocketfft), numpy.interp,
sliding_window_view (== the toeplitz slice, see FCCH_fine_correction), scipy.signal.firwin.
What IS pinned by data in the reference: the 60/30-tap channel filters (tests/golden/*_num.txt,
extracted from gsm_chn_filter_{8x,4x}.fda), the SCH training bits, and every hard-coded constant.
"""
from __future__ import annotations

import math

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from scipy.signal import firwin, lfilter

SYMBOL_RATE = (1625.0 / 6.0) * 1e3  # gsm_sync_demod.m:16

# 64 extended-training-sequence bits of the SCH burst: gsm_SCH_training_sequence_gen.m:17-19 (data)
SCH_TRAINING_BITS = np.array(
    [1, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0,
     0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 0, 0,
     1, 0, 1, 0, 1, 1, 1, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0, 1, 1], dtype=np.int64)


class MatlabIndexError(IndexError):
    """Raised where MATLAB would stop with 'index out of bounds' (SURVEY 8a pitfall 12)."""


def matlab_round(x):
    """MATLAB round(): half away from zero (FCCH_coarse_position.m:35-36 rounds 1562.5 -> 1563)."""
    x = np.asarray(x, dtype=np.float64)
    r = np.sign(x) * np.floor(np.abs(x) + 0.5)
    return float(r) if r.ndim == 0 else r


def fir1(n, wn):
    """fir1(n, Wn): order-n Hamming-window low-pass, unit DC gain (gsm_sync_demod.m:34,
    multi_rtl_sdr_gsm_FCCH_scanner.m:53).  scipy.signal.firwin is the same definition."""
    return firwin(n + 1, wn, window="hamming", pass_zero=True, scale=True)


def _seq_mean(v):
    """mean of a short real vector as sum(v)/n with a left-to-right sum (MATLAB's order for short
    vectors; NumPy's pairwise/unrolled sum would associate differently)."""
    acc = 0.0
    for x in np.asarray(v, dtype=np.float64).ravel():
        acc += float(x)
    return acc / len(v)


def load_num(path):
    """Read a *_num.txt tap fixture (tests/golden)."""
    return np.loadtxt(path, dtype=np.float64, comments="#")


# ------------------------------------------------------------------------------------------------
# a1  raw2iq.m:5-8
# ------------------------------------------------------------------------------------------------
def raw2iq(a):
    """b = raw2iq(a): interleaved byte values (2N x D) -> complex (N x D), per-column mean removed.

    raw2iq.m:6  c = a(1:2:end,:) + 1i.*a(2:2:end,:)
    raw2iq.m:8  b = c - kron(ones(N,1), sum(c,1)./N)
    """
    a = np.asarray(a, dtype=np.float64)
    squeeze = a.ndim == 1
    if squeeze:
        a = a[:, None]
    c = a[0::2, :] + 1j * a[1::2, :]
    # sum(c,1)./size(c,1): MATLAB divides a complex by a REAL scalar part by part.  (NumPy would promote
    # the divisor to complex and multiply by its reciprocal -- a last-bit difference -- so divide explicitly.)
    tot = np.sum(c, axis=0)
    mean = tot.real / c.shape[0] + 1j * (tot.imag / c.shape[0])
    b = c - mean[None, :]
    return b[:, 0] if squeeze else b


# ------------------------------------------------------------------------------------------------
# a2  chn_filter_8x_4x.m:5-15   (and the drivers' inline filter(coef,1,r))
# ------------------------------------------------------------------------------------------------
def matlab_filter(coef, x):
    """filter(coef, 1, x) column-wise, causal, zero initial state (gsm_sync_demod.m:110)."""
    return lfilter(np.asarray(coef, dtype=np.float64), [1.0], np.asarray(x), axis=0)


def chn_filter_8x_4x(s, num):
    """r = chn_filter_8x_4x(s): chn_filter_8x_4x.m:13 filter(Num,1,s); :15 r(1:2:end,:).
    `num` is the 60-tap numerator the reference loads from gsm_chn_filter_8x.mat (:9-10)."""
    r = matlab_filter(num, s)
    return r[0::2]


def chn_filter_4x(s, num):
    """r = chn_filter_4x(s): chn_filter_4x.m:13 filter(Num,1,s), no decimation.  `num` is the 30-tap numerator the
    reference loads from gsm_chn_filter_4x.mat (:8-9)."""
    return matlab_filter(num, s)


# ------------------------------------------------------------------------------------------------
# shared: per-window SNR of move_fft_snr_runtime_avg.m:18-27 / specific_fft_snr_fix_avg.m:11-20
# ------------------------------------------------------------------------------------------------
def _window_snr(P):
    """P: (nwin, fft_len) power spectra -> snr in dB per window.

    [~,max_idx] = max(P); max_set = mod((max_idx+(-1:1))-1, fft_len)+1;
    signal = sum(P(max_set)); noise = sum(P) - signal; snr = 10*log10(signal/noise)
    """
    nwin, fft_len = P.shape
    mi = np.argmax(P, axis=1)  # first max
    rows = np.arange(nwin)
    # sum(chn_tmp(max_set)) adds in the order idx-1, idx, idx+1
    sig = P[rows, (mi - 1) % fft_len] + P[rows, mi]
    sig = sig + P[rows, (mi + 1) % fft_len]
    # sum(chn_tmp): MATLAB's sum is sequential for short vectors
    tot = np.zeros(nwin)
    for k in range(fft_len):
        tot = tot + P[:, k]
    noise = tot - sig
    with np.errstate(divide="ignore", invalid="ignore"):
        return 10.0 * np.log10(sig / noise)


def _power_spectra(s, first, last, fft_len):
    """abs(fft(s(i:i+fft_len-1), fft_len)).^2 for 1-based window starts first..last."""
    s = np.asarray(s)
    if first < 1 or last + fft_len - 1 > len(s):
        raise MatlabIndexError("window outside the signal")
    w = sliding_window_view(s, fft_len)[first - 1:last]
    return np.abs(np.fft.fft(w, axis=1)) ** 2


# ------------------------------------------------------------------------------------------------
# a3  move_fft_snr_runtime_avg.m:5-50
# ------------------------------------------------------------------------------------------------
def move_fft_snr_runtime_avg(s, mv_len, fft_len, th):
    """[hit_flag, hit_idx, hit_avg_snr, hit_snr] = move_fft_snr_runtime_avg(s, mv_len, fft_len, th)

    The per-window spectra are computed for all windows at once (identical values to the reference's
    per-iteration fft); the moving-average / threshold recurrence of :30-42 stays serial, with the
    incrementally updated running sum seeded with 999*mv_len (:11-12, :37-38).
    """
    s = np.asarray(s).ravel()
    nwin = len(s) - (fft_len - 1)
    hit_flag, hit_idx, hit_avg_snr, hit_snr = False, -1, math.inf, math.inf
    if nwin < 1:
        return hit_flag, hit_idx, hit_avg_snr, hit_snr
    snr_all = _window_snr(_power_spectra(s, 1, nwin, fft_len))

    store = [999.0] * mv_len  # store(1) is the newest, store(end) the oldest
    sum_snr = 0.0
    for v in store:  # sum(store_for_moving_avg)
        sum_snr += v
    head = 0  # circular: store_for_moving_avg(end) lives at `head`
    for i in range(nwin):
        snr = float(snr_all[i])
        peak_to_avg = snr - (sum_snr / mv_len)
        if peak_to_avg > th:
            return True, i + 1, snr - peak_to_avg, snr  # :45-49
        sum_snr = sum_snr - store[head]  # :37  minus the oldest
        sum_snr = sum_snr + snr          # :38
        store[head] = snr                # :40-41 shift in (oldest slot is overwritten)
        head = (head + 1) % mv_len
    return hit_flag, hit_idx, hit_avg_snr, hit_snr


# ------------------------------------------------------------------------------------------------
# a4  specific_fft_snr_fix_avg.m:5-34
# ------------------------------------------------------------------------------------------------
def specific_fft_snr_fix_avg(s, target_set, fft_len, th, avg_snr):
    """[hit_flag, hit_idx, hit_snr] = specific_fft_snr_fix_avg(s, target_set, fft_len, th, avg_snr)"""
    s = np.asarray(s).ravel()
    lo, hi = int(target_set[0]), int(target_set[1])
    if hi < lo:
        return False, -1, math.inf
    # the reference indexes window by window (:10-11): a hit in a window that fits returns before a later window would run
    # past the end of s -- only a miss up to there is MATLAB's index error
    if lo < 1:
        raise MatlabIndexError("window outside the signal")
    hi_fit = min(hi, len(s) - (fft_len - 1))
    if hi_fit >= lo:
        snr_all = _window_snr(_power_spectra(s, lo, hi_fit, fft_len))
        for k, i in enumerate(range(lo, hi_fit + 1)):
            snr = float(snr_all[k])
            if snr - avg_snr > th:
                return True, i, snr
    if hi_fit < hi:
        raise MatlabIndexError("window outside the signal")
    return False, -1, math.inf


# ------------------------------------------------------------------------------------------------
# a5  FCCH_coarse_position.m:5-94
# ------------------------------------------------------------------------------------------------
def FCCH_coarse_position(s, decimation_ratio):
    """[position, snr] = FCCH_coarse_position(s, decimation_ratio)

    Returns (position, snr) as float64 row vectors (1x-symbol units, 1-based), or (-1.0, -1.0)
    scalars when no FCCH is found (:7-8, :27-30)."""
    s = np.asarray(s).ravel()
    num_sym_per_slot = 625.0 / 4.0
    num_slot_per_frame = 8
    num_sym_per_frame = num_sym_per_slot * num_slot_per_frame
    len_FCCH_CW = 148
    fft_len = int(2 ** math.floor(math.log2(len_FCCH_CW / decimation_ratio)))  # :17
    length = len(s)
    th = 10.0
    mv_len = 10 * fft_len

    n_first = int(math.ceil(23 * num_sym_per_frame / decimation_ratio))  # :25
    if n_first > length:
        raise MatlabIndexError("s(1:ceil(23 frames)) exceeds the signal")
    hit_flag, hit_idx, hit_avg_snr, hit_snr = move_fft_snr_runtime_avg(s[:n_first], mv_len, fft_len, th)
    if not hit_flag:
        return -1.0, -1.0

    num_sym_between_FCCH = 10 * num_slot_per_frame * num_sym_per_slot
    num_sym_between_FCCH1 = 11 * num_slot_per_frame * num_sym_per_slot
    d0 = int(matlab_round(num_sym_between_FCCH / decimation_ratio))   # :35
    d1 = int(matlab_round(num_sym_between_FCCH1 / decimation_ratio))  # :36

    position = [hit_idx]
    snr = [hit_snr]
    max_offset = 5
    limit = (length - (fft_len - 1)) - max_offset
    while True:
        nxt = position[-1] + d0
        if nxt > limit:  # :49
            break
        hf, hi_, hs = specific_fft_snr_fix_avg(s, (nxt - max_offset, nxt + max_offset), fft_len, th, hit_avg_snr)
        if hf:
            position.append(hi_)
            snr.append(hs)
        else:
            nxt = position[-1] + d1
            if nxt > limit:  # :67
                break
            hf, hi_, hs = specific_fft_snr_fix_avg(s, (nxt - max_offset, nxt + max_offset), fft_len, th, hit_avg_snr)
            if hf:
                position.append(hi_)
                snr.append(hs)
            else:
                break
    position = (np.asarray(position, dtype=np.float64) - 1.0) * decimation_ratio + 1.0  # :91
    return position, np.asarray(snr, dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# shared tone estimator: FCCH_fine_correction.m:143-155 == carrier_correct_post_SCH.m:58-72
# ------------------------------------------------------------------------------------------------
def _fcch_tone_estimate(r, pos, fft_len, sampling_rate):
    """Returns (fcch_mat [fft_len x K], int_phase_rotate [K], phase_rotate [K], fo [K])."""
    K = len(pos)
    fcch_mat = np.zeros((fft_len, K), dtype=np.complex128)
    for i in range(K):
        sp = int(pos[i])
        if sp < 1 or sp + fft_len - 1 > len(r):
            raise MatlabIndexError("FCCH burst window outside the signal")
        fcch_mat[:, i] = r[sp - 1:sp - 1 + fft_len]
    fd = np.abs(np.fft.fft(fcch_mat, axis=0)) ** 2
    fd = np.concatenate([fd[fft_len // 2:, :], fd[:fft_len // 2, :]], axis=0)  # fftshift
    max_idx = np.argmax(fd, axis=0) + 1  # 1-based, first max
    int_phase_rotate = 2.0 * np.pi * (max_idx - ((fft_len / 2) + 1)) / fft_len
    n = np.arange(fft_len, dtype=np.float64)[:, None]
    fcch_mat = fcch_mat * np.exp(-1j * (n * int_phase_rotate[None, :]))
    ang = np.angle(fcch_mat)
    pr = np.exp(1j * ang[1:, :]) / np.exp(1j * ang[:-1, :])
    tot = np.sum(pr, axis=0)   # mean(.,1) = sum ./ n, complex ./ real part by part
    phase_rotate = np.arctan2(tot.imag / pr.shape[0], tot.real / pr.shape[0])
    fo = sampling_rate * (int_phase_rotate + phase_rotate) / (2.0 * np.pi)
    return fcch_mat, int_phase_rotate, phase_rotate, fo


def _interp1_linear(s, xq):
    """interp1((0:len-1)', s, xq, 'linear') for complex s: re and im interpolated independently."""
    xp = np.arange(len(s), dtype=np.float64)
    return np.interp(xq, xp, s.real) + 1j * np.interp(xq, xp, s.imag)


# ------------------------------------------------------------------------------------------------
# a6  FCCH_fine_correction.m:5-197
# ------------------------------------------------------------------------------------------------
def FCCH_fine_correction(s, base_position, oversampling_ratio, carrier_freq, info=None):
    """[FCCH_pos, r, sampling_ppm, carrier_ppm] = FCCH_fine_correction(s, base_position, ov, fc)

    Sentinels as in the reference: FCCH_pos = -1.0 / r = -1.0 (scalars), ppm = inf.
    `info`, if a dict, receives intermediate values (first-round positions, per-burst fo, SNR)."""
    s = np.asarray(s).ravel()
    r = -1.0
    FCCH_pos = -1.0
    sampling_ppm = math.inf
    carrier_ppm = math.inf
    base_position = np.atleast_1d(np.asarray(base_position, dtype=np.float64))
    if len(base_position) < 5:  # :12
        return FCCH_pos, r, sampling_ppm, carrier_ppm

    symbol_rate = SYMBOL_RATE
    sampling_rate = symbol_rate * oversampling_ratio
    len_FCCH_CW = 148
    fft_len = len_FCCH_CW * oversampling_ratio
    half_noise_len = int(math.ceil((fft_len * 200e3 / sampling_rate) / 2))  # :22

    num_fcch_hit = len(base_position)
    FCCH_pos = np.full(num_fcch_hit, np.inf)
    len_s_ov = len(s)
    len_s = len_s_ov // oversampling_ratio
    max_offset = 64
    last_idx = 0
    for i in range(num_fcch_hit):
        position = int(base_position[i])
        if (position + max_offset) > (len_s - len_FCCH_CW + 1):  # :35
            last_idx = i
            break
        sp = (position - max_offset - 1) * oversampling_ratio + 1
        ep = (position + max_offset - 1) * oversampling_ratio + 1
        length = ep - sp + 1
        if sp < 1:
            raise MatlabIndexError("fine-search window starts before the signal")
        # :48-49  toeplitz(...)(len:end, end:-1:1): column k (1-based) == s(sp+k-1 : sp+k-1+fft_len-1)
        seg = s[sp - 1:ep + fft_len - 1]
        win = sliding_window_view(seg, fft_len)  # (len, fft_len), row k-1 == column k
        fft_peak_val = np.max(np.abs(np.fft.fft(win, axis=1)) ** 2, axis=1)  # :50
        max_idx = int(np.argmax(fft_peak_val)) + 1  # :52
        FCCH_pos[i] = sp + max_idx - 1  # :56/:61 (edge peaks only warn)
        last_idx = i + 1
    FCCH_pos = FCCH_pos[:last_idx]
    if info is not None:
        info["first_round_pos"] = FCCH_pos.copy()

    if last_idx >= 5:  # :69
        r = s
        first_FCCH_pos = FCCH_pos[0]
        diff_seq = np.diff(FCCH_pos)
        num_sym_per_frame = (625.0 / 4.0) * 8
        d_ov = 10 * num_sym_per_frame * oversampling_ratio
        d1_ov = 11 * num_sym_per_frame * oversampling_ratio
        max_ppm = 4000
        max_th = math.floor(d_ov * max_ppm * 1e-6)
        max_th1 = math.floor(d1_ov * max_ppm * 1e-6)
        a = diff_seq - d_ov
        a_logical = np.abs(a) < max_th
        b = diff_seq - d1_ov
        b_logical = np.abs(b) < max_th1
        if (np.sum(a_logical) + np.sum(b_logical)) != last_idx - 1:  # :95
            return -1.0, r, sampling_ppm, carrier_ppm
        expected_distance = np.sum(a_logical * d_ov) + np.sum(b_logical * d1_ov)
        actual_distance = FCCH_pos[-1] - FCCH_pos[0]
        mean_ex_percent = (actual_distance - expected_distance) / expected_distance
        sampling_ppm = mean_ex_percent * 1e6
        if mean_ex_percent >= 0:
            max_len = int(math.floor(len(r) / (1 + mean_ex_percent)))
        else:
            max_len = len(r)
        interp_seq = np.arange(max_len, dtype=np.float64) * (1 + mean_ex_percent)  # :123
        r = _interp1_linear(r, interp_seq)  # :125
        step_size = np.zeros(last_idx - 1)
        step_size[a_logical] = d_ov
        step_size[b_logical] = d1_ov
        FCCH_pos = np.cumsum(np.concatenate([[1.0], step_size]))
        first_FCCH_pos = matlab_round((first_FCCH_pos - 1) / (1 + mean_ex_percent)) + 1  # :132
        FCCH_pos = FCCH_pos + first_FCCH_pos - 1
        if (FCCH_pos[-1] + fft_len - 1) > len(r):  # :135
            FCCH_pos = FCCH_pos[:-1]

    num_fcch = len(FCCH_pos)
    if num_fcch >= 5:  # :142
        fcch_mat, int_pr, pr, fo = _fcch_tone_estimate(r, FCCH_pos, fft_len, sampling_rate)
        target_freq = symbol_rate / 4
        if info is not None:
            info["fo_per_burst"] = fo.copy()
        fo = _seq_mean(fo)
        carrier_ppm = 1e6 * (fo - target_freq) / carrier_freq
        comp_freq = target_freq - fo
        comp_phase_rotate = comp_freq * 2 * np.pi / sampling_rate
        r = r * np.exp(1j * (np.arange(len(r), dtype=np.float64) * comp_phase_rotate))  # :165
        # :185-196 SNR gate
        n = np.arange(fft_len, dtype=np.float64)[:, None]
        fcch_mat = fcch_mat * np.exp(-1j * (n * pr[None, :]))
        fd = np.abs(np.fft.fft(fcch_mat, axis=0)) ** 2
        sig_idx = np.concatenate([np.arange(0, 3), np.arange(fft_len - 2, fft_len)])
        noise_idx = np.concatenate([np.arange(3, half_noise_len),
                                    np.arange(fft_len - half_noise_len, fft_len - 2)])
        signal_power = np.sum(fd[sig_idx, :], axis=0)
        noise_power = np.sum(fd[noise_idx, :], axis=0)
        FCCH_snr = 10.0 * np.log10(signal_power / noise_power)
        if info is not None:
            info["fcch_snr"] = FCCH_snr.copy()
        if np.sum(FCCH_snr < 5) > 0:  # :192
            return -1.0, r, sampling_ppm, carrier_ppm
    return FCCH_pos, r, sampling_ppm, carrier_ppm


# ------------------------------------------------------------------------------------------------
# a7  SCH_corr_rate_correction.m:5-181
# ------------------------------------------------------------------------------------------------
def SCH_corr_rate_correction(s, FCCH_pos, sch_training_sequence, oversampling_ratio, info=None):
    """[pos_info, r, sampling_ppm] = SCH_corr_rate_correction(s, FCCH_pos, sch_ts, ov)

    pos_info: (R,2) float64 (col 0 = 1-based start sample, col 1 = 0 FCCH / 1 SCH / 2 BCCH);
    sentinel: a matrix whose elements are all -1 (the reference's [-1,-1] or -ones(3K,2))."""
    r = -1.0
    pos_info = np.array([[-1.0, -1.0]])
    sampling_ppm = math.inf
    FCCH_pos = np.atleast_1d(np.asarray(FCCH_pos, dtype=np.float64))
    if len(FCCH_pos) < 5:  # :11
        return pos_info, r, sampling_ppm
    s = np.asarray(s).ravel()
    sch = np.asarray(sch_training_sequence).ravel()

    num_sym_per_slot = 625.0 / 4.0
    num_sym_per_slot_ov = num_sym_per_slot * oversampling_ratio
    num_slot_per_frame = 8
    num_sym_per_frame = num_sym_per_slot * num_slot_per_frame
    num_sym_per_frame_ov = num_sym_per_frame * oversampling_ratio
    len_ts_ov = 64 * oversampling_ratio
    len_pre_ts_ov = 42 * oversampling_ratio
    fix_off_ov = int((num_sym_per_frame + 42) * oversampling_ratio)  # :26-27

    num_fcch_hit = len(FCCH_pos)
    SCH_pos = np.full(num_fcch_hit, np.inf)
    pos_info = -1.0 * np.ones((3 * num_fcch_hit, 2))  # :32
    len_s_ov = len(s)
    max_offset = 8 * oversampling_ratio
    conj_ts = np.conj(sch)
    for i in range(num_fcch_hit):
        training_sp = int(FCCH_pos[i]) + fix_off_ov
        if (training_sp + max_offset) > (len_s_ov - len_ts_ov + 1):  # :40
            SCH_pos = SCH_pos[:i]
            break
        sp = training_sp - max_offset
        ep = training_sp + max_offset - 5 * oversampling_ratio
        length = ep - sp + 1
        if sp < 1:
            raise MatlabIndexError("SCH search window starts before the signal")
        win = sliding_window_view(s[sp - 1:ep + len_ts_ov - 1], len_ts_ov)  # (len, 512)
        corr_val = np.abs(win @ conj_ts) ** 2  # :53  (sch_ts') * corr_mat
        max_idx = int(np.argmax(corr_val)) + 1
        SCH_pos[i] = sp + max_idx - 1
        if max_idx == 1 or max_idx == length:  # :59
            if info is not None:   # the reference returns here; keep what was computed so far for the tests
                info["first_round_sch_pos"] = SCH_pos[:i + 1].copy()
                info["sch_edge_abort"] = True
            return np.array([[-1.0, -1.0]]), r, sampling_ppm
    if info is not None:
        info["first_round_sch_pos"] = SCH_pos.copy()

    num_sch = len(SCH_pos)
    if num_sch >= 5:  # :84
        r = s
        first_SCH_pos = SCH_pos[0]
        diff_seq = np.diff(SCH_pos)
        d_ov = 10 * num_sym_per_frame_ov
        d1_ov = 11 * num_sym_per_frame_ov
        max_ppm = 400
        max_th = math.floor(d_ov * max_ppm * 1e-6)
        max_th1 = math.floor(d1_ov * max_ppm * 1e-6)
        a = diff_seq - d_ov
        a_logical = np.abs(a) < max_th
        b = diff_seq - d1_ov
        b_logical = np.abs(b) < max_th1
        if (np.sum(a_logical) + np.sum(b_logical)) != num_sch - 1:  # :106
            return pos_info, r, sampling_ppm
        expected_distance = np.sum(a_logical * d_ov) + np.sum(b_logical * d1_ov)
        actual_distance = SCH_pos[-1] - SCH_pos[0]
        mean_ex_percent = (actual_distance - expected_distance) / expected_distance
        sampling_ppm = mean_ex_percent * 1e6
        if mean_ex_percent != 0:  # :120
            if mean_ex_percent > 0:
                max_len = int(math.floor(len(r) / (1 + mean_ex_percent)))
            else:
                max_len = len(r)
            interp_seq = np.arange(max_len, dtype=np.float64) * (1 + mean_ex_percent)
            r = _interp1_linear(r, interp_seq)
        step_size = np.zeros(num_sch - 1)
        step_size[a_logical] = d_ov
        step_size[b_logical] = d1_ov
        SCH_pos = np.cumsum(np.concatenate([[1.0], step_size]))
        first_SCH_pos = matlab_round((first_SCH_pos - 1) / (1 + mean_ex_percent)) + 1
        SCH_pos = SCH_pos + first_SCH_pos - 1

        BCCH_flag = np.zeros(num_sch + 1)
        b_idx = np.nonzero(b_logical)[0] + 1  # 1-based
        BCCH_flag[b_idx + 1 - 1] = 1          # BCCH_flag(b_idx+1) = 1
        bb = b_idx[b_idx >= 5] - 4
        BCCH_flag[bb - 1] = 1                 # BCCH_flag(b_idx(b_idx>=5)-4) = 1

        rows = []
        len_r = len(r)
        for i in range(num_sch):
            sp = SCH_pos[i] - fix_off_ov
            rows.append((sp, 0.0))  # FCCH
            sp = SCH_pos[i] - len_pre_ts_ov
            ep = sp + num_sym_per_slot_ov - 1
            if ep <= len_r:
                rows.append((sp, 1.0))  # SCH
            else:
                break
            sch_sp = sp
            if BCCH_flag[i]:
                runout = False
                for idx in range(1, 5):
                    sp = sch_sp + idx * num_sym_per_frame_ov
                    ep = sp + num_sym_per_slot_ov - 1
                    if ep <= len_r:
                        rows.append((sp, 2.0))  # BCCH
                    else:
                        runout = True
                        break
                if runout:
                    break
        pos_info = np.asarray(rows, dtype=np.float64).reshape(-1, 2)
    return pos_info, r, sampling_ppm


# ------------------------------------------------------------------------------------------------
# a8  carrier_correct_post_SCH.m:5-83
# ------------------------------------------------------------------------------------------------
def carrier_correct_post_SCH(s, pos_info, oversampling_ratio, carrier_freq, info=None):
    """[r, carrier_ppm] = carrier_correct_post_SCH(s, pos_info, ov, fc)"""
    r = -1.0
    carrier_ppm = math.inf
    pos_info = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    if np.all(pos_info == -1):  # :10
        return r, carrier_ppm
    if np.sum(pos_info[:, 1] == 2) < 4:  # :15-19
        return r, carrier_ppm
    s = np.asarray(s).ravel()
    symbol_rate = SYMBOL_RATE
    sampling_rate = symbol_rate * oversampling_ratio
    target_freq = symbol_rate / 4
    fcch_pos = pos_info[pos_info[:, 1] == 0, 0]
    fft_len = 148 * oversampling_ratio
    _, _, _, fo = _fcch_tone_estimate(s, fcch_pos, fft_len, sampling_rate)
    if info is not None:
        info["fo_per_burst"] = fo.copy()
    fo = _seq_mean(fo)
    carrier_ppm = 1e6 * (fo - target_freq) / carrier_freq
    comp_freq = target_freq - fo
    comp_phase_rotate = comp_freq * 2 * np.pi / sampling_rate
    r = s * np.exp(1j * (np.arange(len(s), dtype=np.float64) * comp_phase_rotate))  # :83
    return r, carrier_ppm


# ------------------------------------------------------------------------------------------------
# f4  front end of SCH_demod.m (:53-59, :79-90) -- the equalised SCH bursts; the Viterbi demodulator is out of scope
# ------------------------------------------------------------------------------------------------
def SCH_equalise(s, pos_info, training_sequence, oversampling_ratio):
    """Returns (num_sch, len_fde_ov) complex: x after :90 for every SCH row of pos_info; None when pos_info == -1 (:8-11)."""
    pos_info = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    if np.all(pos_info == -1):  # :8
        return None
    s = np.asarray(s).ravel()
    ts = np.asarray(training_sequence).ravel()
    sch_pos = pos_info[pos_info[:, 1] == 1, 0]  # :13-14
    num_sym_per_slot = 625.0 / 4.0
    num_ef_sym_per_slot = int(matlab_round(num_sym_per_slot - 8.25))  # :22
    len_ts_ov = 64 * oversampling_ratio
    len_pre_ts = 42
    traceback = 30  # :45
    ex_len = 8  # :53
    len_fde_ov = (num_ef_sym_per_slot + 2 * ex_len + traceback) * oversampling_ratio  # :54-55
    sp_t = (ex_len + len_pre_ts) * oversampling_ratio + 1  # :56 (1-based)
    td = np.zeros(len_fde_ov, dtype=np.complex128)
    td[sp_t - 1:sp_t - 1 + len_ts_ov] = ts  # :57-58
    fd_training = np.fft.fft(td)  # :59
    out = np.zeros((len(sch_pos), len_fde_ov), dtype=np.complex128)
    for i, p in enumerate(sch_pos):
        sp = int(p) - ex_len * oversampling_ratio  # :79
        ep = sp + len_fde_ov - 1
        if sp < 1 or ep > len(s):
            raise MatlabIndexError("SCH burst window outside the signal")
        x = s[sp - 1:ep]  # :81
        rt = np.zeros(len_fde_ov, dtype=np.complex128)
        rt[sp_t - 1:sp_t - 1 + len_ts_ov] = x[sp_t - 1:sp_t - 1 + len_ts_ov]  # :83-84
        fd_chn = np.fft.fft(rt) / fd_training  # :85-86
        out[i] = np.fft.ifft(np.fft.fft(x) / fd_chn)  # :88-90
    return out


# ------------------------------------------------------------------------------------------------
# a9  total_ppm_calculation.m:5-21
# ------------------------------------------------------------------------------------------------
def total_ppm_calculation(ppm_in):
    ppm_in = np.atleast_1d(np.asarray(ppm_in, dtype=np.float64))
    if np.all(ppm_in == np.inf):  # :7
        return math.inf
    with np.errstate(invalid="ignore", over="ignore"):
        return float((np.prod(1.0 + ppm_in * 1e-6) - 1.0) * 1e6)


# ------------------------------------------------------------------------------------------------
# driver glue: gsm_sync_demod.m:107-124 (per dongle) and multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,164-185
# ------------------------------------------------------------------------------------------------
def calibrate_stream(raw, coef, sch_training_sequence, carrier_freq,
                     oversampling_ratio=8, coarse_decimation=8, keep_r=False):
    """One dongle of gsm_sync_demod.m:107-124.  raw: uint8-valued interleaved I,Q (2N,).

    Returns a dict with every intermediate the parity tests compare."""
    out = {}
    r = raw2iq(raw)                              # :107
    r = matlab_filter(coef, r)                   # :110
    dec = oversampling_ratio * coarse_decimation
    pos_c, snr_c = FCCH_coarse_position(r[0::dec], coarse_decimation)            # :117
    out["coarse_pos"], out["coarse_snr"] = np.atleast_1d(pos_c), np.atleast_1d(snr_c)
    info = {}
    FCCH_pos, r_c, sp1, cp1 = FCCH_fine_correction(r, pos_c, oversampling_ratio, carrier_freq, info)  # :118
    out["fine_first_round_pos"] = info.get("first_round_pos", np.zeros(0))
    out["fcch_pos"] = np.atleast_1d(np.asarray(FCCH_pos, dtype=np.float64))
    info2 = {}
    pos_info, r_c, sp2 = SCH_corr_rate_correction(r_c, FCCH_pos, sch_training_sequence, oversampling_ratio, info2)  # :119
    out["sch_first_round_pos"] = info2.get("first_round_sch_pos", np.zeros(0))
    out["sch_edge_abort"] = bool(info2.get("sch_edge_abort", False))
    out["pos_info"] = pos_info
    r_c, cp2 = carrier_correct_post_SCH(r_c, pos_info, oversampling_ratio, carrier_freq)  # :120
    out["sampling_ppm"] = np.array([sp1, sp2])
    out["carrier_ppm"] = np.array([cp1, cp2])
    out["total_sampling_ppm"] = total_ppm_calculation([sp1, sp2])  # :123
    out["total_carrier_ppm"] = total_ppm_calculation([cp1, cp2])   # :124
    out["r_len"] = len(r_c) if isinstance(r_c, np.ndarray) else -1
    if keep_r:
        out["r_correct"] = r_c
    return out


def burst_map(pos_info, oversampling_ratio=8):
    """gsm_sync_demod.m:130-134: a = NaN(1, max(round(pos./frame))); a(round(pos(type==k)./frame)) = k for k = 0, 1, 2."""
    pos_info = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    frame = (625.0 / 4.0) * 8 * oversampling_ratio
    idx = matlab_round(pos_info[:, 0] / frame).astype(np.int64)
    a = np.full(int(np.max(idx)), np.nan)
    for k in (0.0, 1.0, 2.0):
        a[idx[pos_info[:, 1] == k] - 1] = k
    return a


def sampling_phase_difference(pos_info_1, pos_info_2, oversampling_ratio=8):
    """gsm_sync_demod.m:151-156 (num_dongle == 2): [num_pos, min_idx] = min(num_pos); the first num_pos burst starts of both
    dongles, their difference pos_tmp2 - pos_tmp1 and the x axis round(pos_tmp(:,1)./frame) of the shorter table
    (the reference indexes that x axis with ALL rows of the shorter table: equal to num_pos by construction)."""
    p1 = np.atleast_2d(np.asarray(pos_info_1, dtype=np.float64))
    p2 = np.atleast_2d(np.asarray(pos_info_2, dtype=np.float64))
    num_pos = [len(p1), len(p2)]
    n = min(num_pos)
    min_idx = int(np.argmin(num_pos))  # first minimum, like MATLAB's min
    frame = (625.0 / 4.0) * 8 * oversampling_ratio
    x = matlab_round((p1, p2)[min_idx][:, 0] / frame)
    return x, p2[:n, 0] - p1[:n, 0]


def scanner_accept(FCCH_pos, FCCH_snr):
    """Acceptance rule of multi_rtl_sdr_gsm_FCCH_scanner.m:168-185 -> (snr, num_hit)."""
    FCCH_pos = np.atleast_1d(np.asarray(FCCH_pos, dtype=np.float64))
    FCCH_snr = np.atleast_1d(np.asarray(FCCH_snr, dtype=np.float64))
    if len(FCCH_pos) >= 3:
        d = np.diff(FCCH_pos)
        a = np.abs(d - 12500) > 50
        if not np.sum(a):
            return float(_seq_mean(FCCH_snr)), float(len(FCCH_pos))
        b = np.abs(d[a] - (12500 + 1250)) > 50
        if not np.sum(b):
            return float(_seq_mean(FCCH_snr)), float(len(FCCH_pos))
    return 0.0, 0.0


def scan_capture(raw, coef, oversampling_ratio=8, coarse_decimation=8):
    """One capture of the scanner: :132-135 front end + :164 detector + :168-185 acceptance."""
    r = matlab_filter(coef, raw2iq(raw))
    dec = oversampling_ratio * coarse_decimation
    pos, snr = FCCH_coarse_position(r[0::dec], coarse_decimation)
    s, n = scanner_accept(pos, snr)
    return {"coarse_pos": np.atleast_1d(pos), "coarse_snr": np.atleast_1d(snr), "snr": s, "num_hit": n}
