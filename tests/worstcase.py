"""Constructed worst cases for the certified shortcuts of the HIP path: inputs on which the reference's
first-max / strict-threshold rules decide by a hair, built with the oracle's own arithmetic so that the oracle's decision
is the designed one (tests/test_worstcase_cpu.py checks that) and the HIP path has to reproduce it (tests/test_gpu_worstcase.py).

  two_tone_stream      FCCH_fine_correction.m:48-52: every fine window holds TWO tone bursts whose aligned spectral peaks
                       differ by a factor 1 + eps -- different bins (one of them outside the 8 bins the certificate
                       evaluates exactly), shifts 64 or more apart (different chunks, edge chunks included), either order.
  coarse_threshold_stream
                       move_fft_snr_runtime_avg.m:30-32 / specific_fft_snr_fix_avg.m:24-26: a decimated stream whose
                       first candidate window (or a hop candidate) has snr - avg - th = delta for a chosen tiny delta.
  degenerate_captures  raw uint8 captures: constant bytes (0/0 -> NaN SNR), rails 0 / 255, a full-scale CW.
Test infrastructure only (imports the oracle)."""
from __future__ import annotations

import math

import numpy as np

from oracle import gsmcal_oracle as o

NFFT = 1184          # 148 * 8
NSHIFT = 1025        # 128 * 8 + 1
WLEN = NFFT + NSHIFT - 1


# ------------------------------------------------------------------------------------------------------------------
# fine search
# ------------------------------------------------------------------------------------------------------------------
def _bin_dot(seg, t, k):
    """X_k(t) = sum_n seg[t+n] exp(-2 pi i k n / NFFT) by a plain dot product"""
    n = np.arange(NFFT)
    return np.dot(seg[t:t + NFFT], np.exp(-2j * np.pi * ((k * n) % NFFT) / NFFT))


def two_tone_window(tA, kA, tB, kB, eps, gA=100.0, noise=None, phases=(0.3, 1.1)):
    """2208 samples holding burst A (bin kA, aligned at shift tA) and burst B (bin kB, aligned at shift tB) with
    |X_kB(tB)|^2 = (1 + eps) |X_kA(tA)|^2 on the summed window (each other's leakage and the noise included)."""
    n = np.arange(NFFT)
    uA = np.zeros(WLEN, complex)
    uB = np.zeros(WLEN, complex)
    uA[tA:tA + NFFT] = np.exp(2j * np.pi * ((kA * n) % NFFT) / NFFT + 1j * phases[0])
    uB[tB:tB + NFFT] = np.exp(2j * np.pi * ((kB * n) % NFFT) / NFFT + 1j * phases[1])
    nz = np.zeros(WLEN, complex) if noise is None else np.asarray(noise, complex)
    aA, aB, aN = _bin_dot(uA, tA, kA), _bin_dot(uB, tA, kA), _bin_dot(nz, tA, kA)
    bA, bB, bN = _bin_dot(uA, tB, kB), _bin_dot(uB, tB, kB), _bin_dot(nz, tB, kB)

    def f(gB):
        return abs(gA * bA + gB * bB + bN) ** 2 - (1.0 + eps) * abs(gA * aA + gB * aB + aN) ** 2

    lo, hi = 0.5 * gA, 2.0 * gA
    assert f(lo) < 0.0 < f(hi)
    for _ in range(200):                                   # bisection down to the last bit of gB
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if f(mid) < 0.0:
            lo = mid
        else:
            hi = mid
    gB = hi if eps > 0 else lo                             # the side on which the sign of f is the designed one
    return gA * uA + gB * uB + nz


def two_tone_stream(cases, gap_symbols=12500, first_symbol=1000, tail=30000, noise_amp=1e-3, seed=7):
    """cases: one (tA, kA, tB, kB, eps) per fine window.  Returns (s, base_position, expected first-round FCCH_pos):
    window i is the search range of coarse position first_symbol + i * gap_symbols (FCCH_fine_correction.m:40-46)."""
    rng = np.random.default_rng(seed)
    n_total = ((first_symbol + gap_symbols * (len(cases) - 1) + 64) * 8 + NFFT + tail)
    s = noise_amp * (rng.standard_normal(n_total) + 1j * rng.standard_normal(n_total))
    base, want = [], []
    for i, (tA, kA, tB, kB, eps) in enumerate(cases):
        p = first_symbol + gap_symbols * i
        sp = (p - 64 - 1) * 8 + 1                           # :40,43 (1-based)
        w0 = sp - 1
        s[w0:w0 + WLEN] = two_tone_window(tA, kA, tB, kB, eps, noise=s[w0:w0 + WLEN].copy(), phases=(0.3 + i, 1.1 + 2 * i))
        base.append(float(p))
        if eps > 0:
            t_win = tB
        elif eps < 0:
            t_win = tA
        else:
            raise ValueError("an exact tie cannot be constructed in floating point")
        want.append(float(sp + t_win))                      # sp + max_idx - 1 with max_idx = t + 1
    return s, np.asarray(base), np.asarray(want)


# (tA, kA, tB, kB): burst A aligned at shift tA in bin kA, burst B at shift tB in bin kB.  The certificate evaluates the 8 bins
# around the strongest bin of the window's MIDDLE 1184 samples (shifts ~512); the other burst is always outside that set.
TWO_TONE_GEOMETRIES = {
    # both maxima in interior chunks, 64 or more shifts apart, bins far apart / 9 apart (just outside the set) / negative frequency
    "interior": [(300, 37, 700, 200), (700, 37, 300, 200), (100, 37, 500, 400), (520, 600, 80, 37), (448, 37, 512, 46),
                 (600, 1150, 200, 37)],
    # one maximum in the first / last 64-shift chunk (the chunks the certificate most often leaves open), shift 0 and 1024 included
    "edge": [(20, 400, 600, 37), (600, 37, 10, 420), (1015, 37, 400, 300), (400, 37, 1020, 300), (0, 37, 640, 90), (500, 37, 1024, 1100)],
}


def fine_peak_map(s, p):
    """the reference's fft_peak_val of coarse position p and the winning bin per shift (FCCH_fine_correction.m:48-50)"""
    from numpy.lib.stride_tricks import sliding_window_view
    sp = (p - 64 - 1) * 8 + 1
    seg = s[sp - 1: sp - 1 + WLEN]
    P = np.abs(np.fft.fft(sliding_window_view(seg, NFFT), axis=1)) ** 2
    return P.max(axis=1), P.argmax(axis=1)


# ------------------------------------------------------------------------------------------------------------------
# coarse detector
# ------------------------------------------------------------------------------------------------------------------
def snr_series(s, fft_len=16):
    s = np.asarray(s).ravel()
    nwin = len(s) - (fft_len - 1)
    return o._window_snr(o._power_spectra(s, 1, nwin, fft_len))


def running_avg_at(snr_all, i, mv_len=160):
    """sum_snr / mv_len as window i (0-based) of move_fft_snr_runtime_avg.m sees it when no earlier window hit: the
    reference's incrementally rounded running sum (:11-12, :37-38)."""
    store = [999.0] * mv_len
    sum_snr = 0.0
    for v in store:
        sum_snr += v
    head = 0
    for j in range(i):
        v = float(snr_all[j])
        sum_snr = sum_snr - store[head]
        sum_snr = sum_snr + v
        store[head] = v
        head = (head + 1) % mv_len
    return sum_snr / mv_len


def _tone16(k=3, phase=0.4):
    return np.exp(2j * np.pi * k * np.arange(16) / 16.0 + 1j * phase)


def _quiet_background(n, seed):
    """complex noise whose 16-point windows, over the moving search's range, never come near the hit threshold on their own"""
    for sd in range(seed, seed + 500):
        rng = np.random.default_rng(sd)
        s = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        snr = snr_series(s[:3594])
        # the running average of noise windows sits near -5.5 dB: keep every window well clear of avg + 10
        if np.max(snr[160:]) < 2.5:
            return s, sd
    raise RuntimeError("no quiet background found")


def _solve(f, lo, hi, delta):
    """(lo, hi): adjacent doubles with f(lo) < delta <= f(hi) (f increasing on the bracket)"""
    assert f(lo) < delta < f(hi), (f(lo), f(hi), delta)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if f(mid) < delta:
            lo = mid
        else:
            hi = mid
    return lo, hi


D0, D1 = 1563, 1719                                         # round(12500/8), round(13750/8): FCCH_coarse_position.m:35-36


def coarse_threshold_stream(delta, where="first", n=15938, seed=11, th=10.0):
    """A decimated detector input (FCCH_coarse_position's s, decimation ratio 8) on which ONE decision of the reference sits
    within |delta| of its threshold, on delta's side of it:
       where="first": window i1 of the moving search has snr - avg - th = margin (every earlier window misses clearly), and a
                      strong burst 700 windows later catches the search when i1 misses      (move_fft_snr_runtime_avg.m:30-32);
       where="hop":   a clear first hit, and the centre candidate of the first +10-frame hop has snr - hit_avg_snr - th = margin
                      (the five candidates before it miss clearly); a strong burst on the +11-frame candidates catches the walk
                      when it misses                                                       (specific_fft_snr_fix_avg.m:24-26).
    Strong bursts on the later +10-frame positions give the walk something to find.  Returns (s, info); seeds are tried
    until the construction holds in the oracle's arithmetic."""
    strong = 40.0
    tone = _tone16()
    for sd in range(seed, seed + 4000, 37):
        s, _ = _quiet_background(n, sd)

        def add(sig, w, g):
            if 0 <= w and w + 16 <= len(sig):
                sig[w:w + 16] += g * tone

        info = {"seed": sd}
        if where == "first":
            i1, i2 = 1000, 1700
            base = s.copy()
            add(base, i2, strong)
            for k in range(1, 10):
                add(base, i1 + D0 * k, strong)
                add(base, i2 - 12 + D0 * k, strong)

            def f(g):
                t = base[: i1 + 16].copy()
                add(t, i1, g)
                sn = snr_series(t)
                return float(sn[i1]) - running_avg_at(sn, i1) - th

            lo, hi = _solve(f, 0.5, 30.0, delta)
            g = hi if delta > 0 else lo
            add(base, i1, g)
            sn = snr_series(base[: i1 + 16])
            margins = np.array([float(sn[j]) - running_avg_at(sn, j) - th for j in range(i1 - 20, i1 + 1)])
            info["margin"] = float(margins[-1])
            ok = np.all(margins[:-1] < -1e-3) and np.sign(margins[-1]) == np.sign(delta) and abs(margins[-1]) <= 4.0 * abs(delta)
            hf, hidx, _, _ = o.move_fft_snr_runtime_avg(base[:3594], 160, 16, th)
            ok = ok and hf and ((hidx == i1 + 1) if delta > 0 else (hidx > i1 + 1))
            if ok:
                return base, info
        else:
            p = 1200
            base = s.copy()
            add(base, p, strong)
            hf, hidx, hit_avg, _ = o.move_fft_snr_runtime_avg(base[:3594], 160, 16, th)
            if not hf or not (p - 15 <= hidx - 1 <= p):
                continue
            h = hidx - 1                                    # 0-based window of the first hit
            q, q1 = h + D0, h + D1                          # centre candidates of the two hop branches (0-based windows)
            add(base, q1, strong)
            for k in range(1, 9):
                add(base, q + D0 * k, strong)
                add(base, q1 - 8 + D0 * k, strong)

            def f(g):
                t = base[q - 20: q + 40].copy()
                t[20:36] += g * tone
                return float(snr_series(t)[20]) - hit_avg - th

            lo, hi = _solve(f, 0.5, 30.0, delta)
            g = hi if delta > 0 else lo
            add(base, q, g)
            sn = snr_series(base[q - 5: q + 5 + 16])         # the 11 candidates of the +10-frame branch
            margins = sn - hit_avg - th
            info["margin"] = float(margins[5])
            info["first_hit"] = hidx
            ok = np.all(margins[:5] < -1e-3) and np.sign(margins[5]) == np.sign(delta) and abs(margins[5]) <= 4.0 * abs(delta)
            if delta < 0:
                ok = ok and np.all(margins[6:] < -1e-3)       # the whole branch misses: the walk must try +11 frames
            if ok:
                return base, info
    raise RuntimeError("construction failed for every seed tried")


# ------------------------------------------------------------------------------------------------------------------
# degenerate captures
# ------------------------------------------------------------------------------------------------------------------
def degenerate_captures(num_frames=102):
    n = num_frames * 10000
    k = np.arange(n)
    caps = {}
    caps["constant_128"] = np.full(2 * n, 128, np.uint8)                       # raw2iq -> all zeros: every SNR is 0/0
    caps["all_zero"] = np.zeros(2 * n, np.uint8)
    caps["all_255"] = np.full(2 * n, 255, np.uint8)
    sq = np.where((k // 4) % 2 == 0, 255, 0).astype(np.uint8)                   # rail to rail, fs/8 square wave on I, inverted on Q
    caps["rail_to_rail"] = np.stack([sq, 255 - sq], axis=1).reshape(-1)
    ph = 2.0 * np.pi * (67708.33 / (270833.333 * 8)) * k                        # a full-scale CW at the FCCH offset, never off
    cw = np.stack([127.5 + 127.5 * np.cos(ph), 127.5 + 127.5 * np.sin(ph)], axis=1)
    caps["full_scale_cw"] = np.clip(np.floor(cw + 0.5), 0, 255).astype(np.uint8).reshape(-1)
    rng = np.random.default_rng(3)
    hard = np.where(rng.standard_normal(2 * n) > 0, 255, 0).astype(np.uint8)    # hard-clipped noise: only the rails occur
    caps["clipped_noise"] = hard
    return caps


def constant_stretch_captures(raw, starts=(200 * 64, 3000 * 64), length=40 * 64, level=128):
    """Captures with a stretch of dropped samples filled by a constant.
      "near_mean": the stretch holds the capture's DC level rounded to a byte -- after raw2iq a small non-zero constant: the
                   16-point windows inside it have ONE non-zero bin and exactly zero noise, SNR = +inf in the reference;
      "exact_mean": the capture is dithered (+1 LSB on a random subset of the samples outside the stretch) until its I and Q
                   means are EXACTLY `level`, and the stretch holds `level`: raw2iq gives exact zeros there, every window
                   SNR is 0/0 = NaN, and the NaN stays in the reference's running sum for good (move_fft_snr_runtime_avg.m:37-41).
    Returns {name: uint8 capture}."""
    raw = np.asarray(raw, dtype=np.uint8)
    n = len(raw) // 2
    out = {}
    rng = np.random.default_rng(5)
    for lo in starts:
        c = raw.copy()
        for comp in (0, 1):
            c[2 * lo + comp: 2 * (lo + length): 2] = int(round(float(np.mean(raw[comp::2]))))
        out[f"near_mean_{lo // 64}"] = c
        e = raw.copy()
        for comp in (0, 1):
            v = e[comp::2].astype(np.int64)
            v[lo: lo + length] = level
            need = level * n - int(v.sum())                  # +1 (or -1) on this many samples outside the stretch
            idx = np.concatenate([np.arange(0, lo), np.arange(lo + length, n)])
            ok = idx[(v[idx] + np.sign(need) >= 0) & (v[idx] + np.sign(need) <= 255)]
            pick = rng.choice(ok, size=abs(need), replace=False)
            v[pick] += int(np.sign(need))
            assert v.sum() == level * n
            e[comp::2] = v.astype(np.uint8)
        out[f"exact_mean_{lo // 64}"] = e
    return out
