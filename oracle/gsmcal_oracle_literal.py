"""TEST INFRASTRUCTURE ONLY -- a SECOND, deliberately literal restatement of the reference's hot path.

oracle/gsmcal_oracle.py restates the nine .m files with vectorised NumPy (sliding_window_view for the toeplitz slices,
batched FFTs, np.interp).  A misreading of a .m file shared by the oracle and the HIP kernels (written by the same
author) would pass every HIP <-> oracle test.  This module restates the same .m files a second time, as literally as
the language allows and sharing NO helper with the first: one statement per MATLAB statement, scalar loops where MATLAB
loops, an explicit toeplitz() followed by the reference's (len:end, end:-1:1) slice, the array shift of the moving
average, MATLAB's 1-based indices kept in the arithmetic (`m1()` converts at the last moment), DFTs from the definition
where the sizes allow, and interp1 written out from its documented two-point formula.  tests/test_oracle_cpu.py
cross-checks the two on seeded streams.

PARITY STAYS UNPINNED: this tightens the oracle against slips of one restatement; it is still not MATLAB output.
Citations are relative to the reference repo root.
"""
from __future__ import annotations

import cmath
import math

import numpy as np
from scipy.linalg import toeplitz

SYMBOL_RATE = (1625.0 / 6.0) * 1e3


def m_round(x):
    """MATLAB round: ties away from zero."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def dft_definition(x):
    """fft(x) by its definition, X(k) = sum_n x(n) exp(-2 pi i (k-1)(n-1)/N), for short vectors."""
    n = len(x)
    k = np.arange(n)
    w = np.exp(-2j * np.pi * np.outer(k, k) / n)
    return w @ np.asarray(x, dtype=np.complex128)


def fft_cols(m):
    """fft(M, [], 1): definition DFT for small column lengths, library FFT otherwise."""
    m = np.asarray(m, dtype=np.complex128)
    if m.shape[0] <= 64:
        return np.stack([dft_definition(m[:, j]) for j in range(m.shape[1])], axis=1)
    return np.fft.fft(m, axis=0)


# ---- raw2iq.m -------------------------------------------------------------------------------------------------------
def raw2iq(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    rows, cols = a.shape
    c = np.zeros((rows // 2, cols), dtype=np.complex128)
    for j in range(cols):
        for i in range(rows // 2):
            c[i, j] = complex(a[2 * i, j], a[2 * i + 1, j])          # :6  a(1:2:end,:) + 1i.*a(2:2:end,:)
    b = np.zeros_like(c)
    for j in range(cols):
        sr = 0.0
        si = 0.0
        for i in range(rows // 2):                                   # :8  sum(c,1)
            sr += c[i, j].real
            si += c[i, j].imag
        mean = complex(sr / (rows // 2), si / (rows // 2))           #     ./size(c,1)
        for i in range(rows // 2):
            b[i, j] = c[i, j] - mean
    return b


# ---- filter(coef,1,x) (gsm_sync_demod.m:110) -----------------------------------------------------------------------
def filter_fir(coef, x):
    """y(n) = sum_k coef(k) x(n-k+1), zero initial state -- the difference equation itself, column by column."""
    x = np.asarray(x, dtype=np.complex128)
    squeeze = x.ndim == 1
    if squeeze:
        x = x.reshape(-1, 1)
    y = np.zeros_like(x)
    for k, c in enumerate(coef):
        y[k:, :] += c * x[: x.shape[0] - k, :]
    return y[:, 0] if squeeze else y


# ---- move_fft_snr_runtime_avg.m -------------------------------------------------------------------------------------
def move_fft_snr_runtime_avg(s, mv_len, fft_len, th):
    hit_flag = False
    hit_idx = -1
    hit_avg_snr = math.inf
    hit_snr = math.inf
    store = [999.0] * mv_len                                          # :11
    sum_snr = 0.0
    for v in store:                                                   # :12
        sum_snr += v
    length = len(s)
    snr = peak_to_avg = 0.0
    i = 0
    for i in range(1, length - (fft_len - 1) + 1):                    # :17  (1-based i)
        chn_tmp = s[i - 1:i - 1 + fft_len]                            # :18
        spec = dft_definition(chn_tmp)                                # :19
        chn_tmp = [abs(z) ** 2 for z in spec]
        max_idx = 1
        for k in range(2, fft_len + 1):                               # :22 first maximum
            if chn_tmp[k - 1] > chn_tmp[max_idx - 1]:
                max_idx = k
        max_set = [((max_idx + d) - 1) % fft_len + 1 for d in (-1, 0, 1)]   # :23
        signal_power = 0.0
        for k in max_set:                                             # :24
            signal_power += chn_tmp[k - 1]
        total = 0.0
        for v in chn_tmp:                                             # :26
            total += v
        noise_power = total - signal_power
        snr = 10.0 * math.log10(signal_power / noise_power) if noise_power > 0 and signal_power > 0 else \
            float(10.0 * np.log10(np.float64(signal_power) / np.float64(noise_power)))
        peak_to_avg = snr - (sum_snr / mv_len)                        # :30
        if peak_to_avg > th:                                          # :32
            hit_flag = True
            break
        sum_snr = sum_snr - store[-1]                                 # :37
        sum_snr = sum_snr + snr                                       # :38
        store[1:] = store[:-1]                                        # :40
        store[0] = snr                                                # :41
    if hit_flag:
        hit_idx = i
        hit_snr = snr
        hit_avg_snr = snr - peak_to_avg
    return hit_flag, hit_idx, hit_avg_snr, hit_snr


# ---- specific_fft_snr_fix_avg.m -------------------------------------------------------------------------------------
def specific_fft_snr_fix_avg(s, target_set, fft_len, th, avg_snr):
    hit_flag = False
    hit_idx = -1
    hit_snr = math.inf
    for i in range(int(target_set[0]), int(target_set[1]) + 1):
        if i < 1 or i + fft_len - 1 > len(s):
            raise IndexError("index exceeds matrix dimensions")
        spec = dft_definition(s[i - 1:i - 1 + fft_len])
        chn_tmp = [abs(z) ** 2 for z in spec]
        max_idx = 1
        for k in range(2, fft_len + 1):
            if chn_tmp[k - 1] > chn_tmp[max_idx - 1]:
                max_idx = k
        signal_power = 0.0
        for d in (-1, 0, 1):
            signal_power += chn_tmp[((max_idx + d) - 1) % fft_len]
        total = 0.0
        for v in chn_tmp:
            total += v
        snr = float(10.0 * np.log10(np.float64(signal_power) / np.float64(total - signal_power)))
        if snr - avg_snr > th:
            hit_flag = True
            hit_idx = i
            hit_snr = snr
            break
    return hit_flag, hit_idx, hit_snr


# ---- FCCH_coarse_position.m -----------------------------------------------------------------------------------------
def FCCH_coarse_position(s, decimation_ratio):
    position = -1.0
    snr = -1.0
    num_sym_per_slot = 625 / 4
    num_slot_per_frame = 8
    num_sym_per_frame = num_sym_per_slot * num_slot_per_frame
    len_FCCH_CW = 148
    fft_len = 2 ** math.floor(math.log2(len_FCCH_CW / decimation_ratio))            # :17
    length = len(s)
    th = 10
    mv_len = 10 * fft_len
    n_first = math.ceil(23 * num_sym_per_frame / decimation_ratio)
    if n_first > length:
        raise IndexError("index exceeds matrix dimensions")
    hit_flag, hit_idx, hit_avg_snr, hit_snr = move_fft_snr_runtime_avg(s[:n_first], mv_len, fft_len, th)   # :25
    if not hit_flag:                                                                # :27
        return position, snr
    num_sym_between_FCCH = 10 * num_slot_per_frame * num_sym_per_slot
    num_sym_between_FCCH1 = 11 * num_slot_per_frame * num_sym_per_slot
    d0 = m_round(num_sym_between_FCCH / decimation_ratio)                           # :35
    d1 = m_round(num_sym_between_FCCH1 / decimation_ratio)                          # :36
    max_num_fcch = math.ceil(length / (num_sym_between_FCCH / decimation_ratio))    # :38
    position = [0.0] * max_num_fcch
    snr = [0.0] * max_num_fcch
    position[0] = hit_idx
    snr[0] = hit_snr
    set_idx = 1
    max_offset = 5
    while True:                                                                     # :46
        next_position = position[set_idx - 1] + d0
        if next_position > (length - (fft_len - 1)) - max_offset:                   # :49
            break
        hf, hi, hs = specific_fft_snr_fix_avg(s, (next_position - max_offset, next_position + max_offset), fft_len, th, hit_avg_snr)
        if hf:
            set_idx += 1
            if set_idx > len(position):
                position.append(0.0)
                snr.append(0.0)
            position[set_idx - 1] = hi
            snr[set_idx - 1] = hs
        else:
            next_position = position[set_idx - 1] + d1                              # :65
            if next_position > (length - (fft_len - 1)) - max_offset:
                break
            hf, hi, hs = specific_fft_snr_fix_avg(s, (next_position - max_offset, next_position + max_offset), fft_len, th, hit_avg_snr)
            if hf:
                set_idx += 1
                if set_idx > len(position):
                    position.append(0.0)
                    snr.append(0.0)
                position[set_idx - 1] = hi
                snr[set_idx - 1] = hs
            else:
                break
    position = [(p - 1) * decimation_ratio + 1 for p in position[:set_idx]]         # :89-91
    return np.array(position, dtype=np.float64), np.array(snr[:set_idx], dtype=np.float64)


# ---- interp1(x, v, xq, 'linear') with x = 0:len-1 -------------------------------------------------------------------
def interp1_linear_unit_grid(v, xq):
    """The documented two-point formula vq = v(k) + (xq - x(k)) (v(k+1) - v(k)) / (x(k+1) - x(k)) on the grid x = 0:len-1,
    written with the weights (1-t) v0 + t v1 -- NOT the v0 + t (v1 - v0) form the first oracle's np.interp and the kernels
    use, so a formula-dependent difference would show (it stays at the 1e-16 level)."""
    v = np.asarray(v, dtype=np.complex128)
    out = np.zeros(len(xq), dtype=np.complex128)
    last = len(v) - 1
    for j, q in enumerate(xq):
        k = int(math.floor(q))
        if k >= last:
            if q > last:
                out[j] = complex(math.nan, math.nan)
                continue
            out[j] = v[last]
            continue
        t = q - k
        out[j] = (1.0 - t) * v[k] + t * v[k + 1]
    return out


def _tone_estimate(r, pos_list, fft_len, sampling_rate):
    """FCCH_fine_correction.m:143-155 / carrier_correct_post_SCH.m:58-72."""
    num = len(pos_list)
    fcch_mat = np.zeros((fft_len, num), dtype=np.complex128)
    for i in range(num):
        sp = int(pos_list[i])
        if sp < 1 or sp + fft_len - 1 > len(r):
            raise IndexError("index exceeds matrix dimensions")
        fcch_mat[:, i] = r[sp - 1:sp - 1 + fft_len]
    fd = np.abs(fft_cols(fcch_mat)) ** 2
    fd = np.vstack([fd[fft_len // 2:, :], fd[:fft_len // 2, :]])      # :149
    int_phase_rotate = np.zeros(num)
    for i in range(num):
        max_idx = 1
        for k in range(2, fft_len + 1):
            if fd[k - 1, i] > fd[max_idx - 1, i]:
                max_idx = k
        int_phase_rotate[i] = 2.0 * math.pi * (max_idx - ((fft_len / 2) + 1)) / fft_len
    phase_rotate = np.zeros(num)
    fo = np.zeros(num)
    for i in range(num):
        col = [fcch_mat[n, i] * cmath.exp(-1j * (n * int_phase_rotate[i])) for n in range(fft_len)]
        fcch_mat[:, i] = col
        acc = 0j
        for n in range(fft_len - 1):
            acc += cmath.exp(1j * cmath.phase(col[n + 1])) / cmath.exp(1j * cmath.phase(col[n]))
        acc = complex(acc.real / (fft_len - 1), acc.imag / (fft_len - 1))
        phase_rotate[i] = cmath.phase(acc)
        fo[i] = sampling_rate * (int_phase_rotate[i] + phase_rotate[i]) / (2 * math.pi)
    return fcch_mat, int_phase_rotate, phase_rotate, fo


# ---- FCCH_fine_correction.m -----------------------------------------------------------------------------------------
def FCCH_fine_correction(s, base_position, oversampling_ratio, carrier_freq):
    s = np.asarray(s, dtype=np.complex128).ravel()
    r = -1.0
    FCCH_pos = -1.0
    sampling_ppm = math.inf
    carrier_ppm = math.inf
    base_position = np.atleast_1d(np.asarray(base_position, dtype=np.float64))
    if len(base_position) < 5:                                                      # :12
        return FCCH_pos, r, sampling_ppm, carrier_ppm, []
    symbol_rate = SYMBOL_RATE
    sampling_rate = symbol_rate * oversampling_ratio
    len_FCCH_CW = 148
    fft_len = len_FCCH_CW * oversampling_ratio
    half_noise_len = math.ceil((fft_len * 200e3 / sampling_rate) / 2)
    num_fcch_hit = len(base_position)
    FCCH_pos = [math.inf] * num_fcch_hit
    len_s_ov = len(s)
    len_s = math.floor(len_s_ov / oversampling_ratio)
    max_offset = 64
    last_idx = 0
    for i in range(1, num_fcch_hit + 1):
        position = int(base_position[i - 1])
        if (position + max_offset) > (len_s - len_FCCH_CW + 1):                     # :35
            last_idx = i - 1
            break
        sp = position - max_offset
        ep = position + max_offset
        sp = (sp - 1) * oversampling_ratio + 1
        ep = (ep - 1) * oversampling_ratio + 1
        length = ep - sp + 1
        if sp < 1 or ep + fft_len - 1 > len_s_ov:
            raise IndexError("index exceeds matrix dimensions")
        col = s[sp - 1:ep + fft_len - 1]                                            # :48 s(sp:(ep+fft_len-1))
        row = np.concatenate([[s[sp - 1]], np.zeros(length - 1)])
        fft_mat = toeplitz(col, row)
        fft_mat = fft_mat[length - 1:, ::-1]                                        # :49 (len:end, end:-1:1)
        power = np.abs(np.fft.fft(fft_mat, n=fft_len, axis=0)) ** 2                 # :50
        fft_peak_val = [max(power[:, j]) for j in range(length)]
        max_idx = 1
        for j in range(2, length + 1):                                              # :52 first maximum
            if fft_peak_val[j - 1] > fft_peak_val[max_idx - 1]:
                max_idx = j
        FCCH_pos[i - 1] = sp + max_idx - 1
        last_idx = i
    FCCH_pos = FCCH_pos[:last_idx]
    first_round = list(FCCH_pos)
    if last_idx >= 5:
        r = s
        first_FCCH_pos = FCCH_pos[0]
        diff_seq = [FCCH_pos[k + 1] - FCCH_pos[k] for k in range(last_idx - 1)]
        num_sym_per_frame = (625 / 4) * 8
        d_ov = 10 * num_sym_per_frame * oversampling_ratio
        d1_ov = 11 * num_sym_per_frame * oversampling_ratio
        max_ppm = 4000
        max_th = math.floor(d_ov * max_ppm * 1e-6)
        max_th1 = math.floor(d1_ov * max_ppm * 1e-6)
        a_logical = [abs(d - d_ov) < max_th for d in diff_seq]
        b_logical = [abs(d - d1_ov) < max_th1 for d in diff_seq]
        if sum(a_logical) + sum(b_logical) != last_idx - 1:                         # :95
            return -1.0, r, sampling_ppm, carrier_ppm, first_round
        expected = sum(d_ov for f in a_logical if f) + sum(d1_ov for f in b_logical if f)
        actual = FCCH_pos[-1] - FCCH_pos[0]
        e = (actual - expected) / expected
        sampling_ppm = e * 1e6
        max_len = math.floor(len(r) / (1 + e)) if e >= 0 else len(r)
        interp_seq = [k * (1 + e) for k in range(max_len)]                          # :123
        r = interp1_linear_unit_grid(r, interp_seq)                                 # :125
        step = [0.0] * (last_idx - 1)
        for k in range(last_idx - 1):
            if a_logical[k]:
                step[k] = d_ov
            if b_logical[k]:
                step[k] = d1_ov
        grid = [1.0]
        for st in step:                                                             # :131 cumsum([1 step_size])
            grid.append(grid[-1] + st)
        first_FCCH_pos = m_round((first_FCCH_pos - 1) / (1 + e)) + 1                # :132
        FCCH_pos = [g + first_FCCH_pos - 1 for g in grid]
        if FCCH_pos[-1] + fft_len - 1 > len(r):                                     # :135
            FCCH_pos = FCCH_pos[:-1]
    if len(FCCH_pos) >= 5:
        fcch_mat, _ipr, pr, fo = _tone_estimate(r, FCCH_pos, fft_len, sampling_rate)
        target_freq = symbol_rate / 4
        mean_fo = 0.0
        for v in fo:
            mean_fo += v
        mean_fo /= len(fo)
        carrier_ppm = 1e6 * (mean_fo - target_freq) / carrier_freq
        comp_phase_rotate = (target_freq - mean_fo) * 2 * math.pi / sampling_rate
        r = np.array([r[k] * cmath.exp(1j * (k * comp_phase_rotate)) for k in range(len(r))])     # :165
        snr = []
        for i in range(len(FCCH_pos)):
            col = np.array([fcch_mat[n, i] * cmath.exp(-1j * (n * pr[i])) for n in range(fft_len)])   # :185
            fd = np.abs(np.fft.fft(col)) ** 2
            sig_idx = [1, 2, 3, fft_len - 1, fft_len]                               # :187 (1-based)
            noise_idx = list(range(4, half_noise_len + 1)) + list(range(fft_len - half_noise_len + 1, fft_len - 2 + 1))
            sp_ = sum(fd[k - 1] for k in sig_idx)
            npow = sum(fd[k - 1] for k in noise_idx)
            snr.append(10 * math.log10(sp_ / npow))
        if sum(1 for v in snr if v < 5) > 0:                                        # :192
            return -1.0, r, sampling_ppm, carrier_ppm, first_round
    return np.array(FCCH_pos, dtype=np.float64), r, sampling_ppm, carrier_ppm, first_round


# ---- SCH_corr_rate_correction.m -------------------------------------------------------------------------------------
def SCH_corr_rate_correction(s, FCCH_pos, sch_training_sequence, oversampling_ratio):
    r = -1.0
    pos_info = np.array([[-1.0, -1.0]])
    sampling_ppm = math.inf
    FCCH_pos = np.atleast_1d(np.asarray(FCCH_pos, dtype=np.float64))
    if len(FCCH_pos) < 5:                                                           # :11
        return pos_info, r, sampling_ppm
    s = np.asarray(s, dtype=np.complex128).ravel()
    ts = np.asarray(sch_training_sequence, dtype=np.complex128).ravel()
    num_sym_per_slot_ov = (625 / 4) * oversampling_ratio
    num_sym_per_frame = (625 / 4) * 8
    num_sym_per_frame_ov = num_sym_per_frame * oversampling_ratio
    len_ts_ov = 64 * oversampling_ratio
    len_pre_ts_ov = 42 * oversampling_ratio
    fix_off_ov = (num_sym_per_frame + 42) * oversampling_ratio
    num_fcch_hit = len(FCCH_pos)
    SCH_pos = [math.inf] * num_fcch_hit
    pos_info = -1.0 * np.ones((3 * num_fcch_hit, 2))                                # :32
    len_s_ov = len(s)
    max_offset = 8 * oversampling_ratio
    for i in range(1, num_fcch_hit + 1):
        training_sp = int(FCCH_pos[i - 1] + fix_off_ov)
        if (training_sp + max_offset) > (len_s_ov - len_ts_ov + 1):                 # :40
            SCH_pos = SCH_pos[:i - 1]
            break
        sp = training_sp - max_offset
        ep = training_sp + max_offset - 5 * oversampling_ratio
        length = ep - sp + 1
        if sp < 1:
            raise IndexError("index exceeds matrix dimensions")
        col = s[sp - 1:ep + len_ts_ov - 1]
        row = np.concatenate([[s[sp - 1]], np.zeros(length - 1)])
        corr_mat = toeplitz(col, row)
        corr_mat = corr_mat[length - 1:, ::-1]                                      # :51
        corr_val = np.abs(np.conj(ts) @ corr_mat) ** 2                              # :53  (sch_ts') * corr_mat
        max_idx = 1
        for j in range(2, length + 1):
            if corr_val[j - 1] > corr_val[max_idx - 1]:
                max_idx = j
        SCH_pos[i - 1] = sp + max_idx - 1
        if max_idx == 1 or max_idx == length:                                       # :59
            return np.array([[-1.0, -1.0]]), r, sampling_ppm
    num_sch = len(SCH_pos)
    if num_sch >= 5:
        r = s
        first_SCH_pos = SCH_pos[0]
        diff_seq = [SCH_pos[k + 1] - SCH_pos[k] for k in range(num_sch - 1)]
        d_ov = 10 * num_sym_per_frame_ov
        d1_ov = 11 * num_sym_per_frame_ov
        max_ppm = 400
        max_th = math.floor(d_ov * max_ppm * 1e-6)
        max_th1 = math.floor(d1_ov * max_ppm * 1e-6)
        a_logical = [abs(d - d_ov) < max_th for d in diff_seq]
        b_logical = [abs(d - d1_ov) < max_th1 for d in diff_seq]
        if sum(a_logical) + sum(b_logical) != num_sch - 1:                          # :106
            return pos_info, r, sampling_ppm
        expected = sum(d_ov for f in a_logical if f) + sum(d1_ov for f in b_logical if f)
        actual = SCH_pos[-1] - SCH_pos[0]
        e = (actual - expected) / expected
        sampling_ppm = e * 1e6
        if e != 0:                                                                  # :120
            max_len = math.floor(len(r) / (1 + e)) if e > 0 else len(r)
            r = interp1_linear_unit_grid(r, [k * (1 + e) for k in range(max_len)])
        step = [0.0] * (num_sch - 1)
        for k in range(num_sch - 1):
            if a_logical[k]:
                step[k] = d_ov
            if b_logical[k]:
                step[k] = d1_ov
        grid = [1.0]
        for st in step:
            grid.append(grid[-1] + st)
        first_SCH_pos = m_round((first_SCH_pos - 1) / (1 + e)) + 1
        SCH_pos = [g + first_SCH_pos - 1 for g in grid]
        BCCH_flag = [0] * (num_sch + 1)                                             # :138 (1-based below)
        b_idx = [k + 1 for k in range(num_sch - 1) if b_logical[k]]
        for b in b_idx:
            BCCH_flag[(b + 1) - 1] = 1                                              # :140
        for b in b_idx:
            if b >= 5:
                BCCH_flag[(b - 4) - 1] = 1                                          # :141
        rows = [list(x) for x in pos_info]
        burst_idx = 1

        def put(idx, sp_, kind):
            while idx > len(rows):                                                  # MATLAB grows the matrix on assignment
                rows.append([0.0, 0.0])
            rows[idx - 1] = [sp_, kind]

        for i in range(1, num_sch + 1):
            sp = SCH_pos[i - 1] - fix_off_ov
            put(burst_idx, sp, 0.0)
            burst_idx += 1
            sp = SCH_pos[i - 1] - len_pre_ts_ov
            ep = sp + num_sym_per_slot_ov - 1
            if ep <= len(r):
                put(burst_idx, sp, 1.0)
                burst_idx += 1
            else:
                break
            sch_sp = sp
            if BCCH_flag[i - 1]:
                runout = False
                for idx in range(1, 5):
                    sp = sch_sp + idx * num_sym_per_frame_ov
                    ep = sp + num_sym_per_slot_ov - 1
                    if ep <= len(r):
                        put(burst_idx, sp, 2.0)
                        burst_idx += 1
                    else:
                        runout = True
                        break
                if runout:
                    break
        pos_info = np.array(rows[:burst_idx - 1], dtype=np.float64).reshape(-1, 2)  # :181
    return pos_info, r, sampling_ppm


# ---- carrier_correct_post_SCH.m -------------------------------------------------------------------------------------
def carrier_correct_post_SCH(s, pos_info, oversampling_ratio, carrier_freq):
    r = -1.0
    carrier_ppm = math.inf
    pos_info = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    if all(v == -1 for v in pos_info.ravel()):                                      # :10
        return r, carrier_ppm
    if sum(1 for v in pos_info[:, 1] if v == 2) < 4:                                # :15
        return r, carrier_ppm
    s = np.asarray(s, dtype=np.complex128).ravel()
    symbol_rate = SYMBOL_RATE
    sampling_rate = symbol_rate * oversampling_ratio
    target_freq = symbol_rate / 4
    fcch_pos = [pos_info[k, 0] for k in range(len(pos_info)) if pos_info[k, 1] == 0]
    fft_len = 148 * oversampling_ratio
    _, _, _, fo = _tone_estimate(s, fcch_pos, fft_len, sampling_rate)
    mean_fo = 0.0
    for v in fo:
        mean_fo += v
    mean_fo /= len(fo)
    carrier_ppm = 1e6 * (mean_fo - target_freq) / carrier_freq
    comp_phase_rotate = (target_freq - mean_fo) * 2 * math.pi / sampling_rate
    r = np.array([s[k] * cmath.exp(1j * (k * comp_phase_rotate)) for k in range(len(s))])
    return r, carrier_ppm


# ---- total_ppm_calculation.m ----------------------------------------------------------------------------------------
def total_ppm_calculation(ppm_in):
    ppm_in = list(np.atleast_1d(ppm_in))
    if all(v == math.inf for v in ppm_in):
        return math.inf
    tmp = 1.0
    for v in ppm_in:
        tmp = tmp * (1 + v * 1e-6)
    return (tmp - 1) * 1e6
