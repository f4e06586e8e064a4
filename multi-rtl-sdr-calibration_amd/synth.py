"""Seeded synthetic 8x-oversampled GSM BCCH-carrier IQ (uint8, interleaved I,Q) and the GMSK
modulator that also produces the SCH training-sequence template.

Nothing here is derived from reference code: the reference gets its template from MathWorks'
comm.GMSKModulator (gsm_SCH_training_sequence_gen.m:14,39 -- Communications Toolbox, not in the
repo) and its IQ from RTL-SDR dongles.  What is taken from the reference is data: the 64 SCH
extended-training bits (gsm_SCH_training_sequence_gen.m:17-19), the differential pre-coding rule
d = ~xor(b[i], b[i-1]) with b[-1] = 0 (:32), BT = 0.3, pulse length 4, 8 samples/symbol (:9-14),
and the burst geometry SCH_corr_rate_correction.m:22-27 assumes (training sequence 42 symbols
into the SCH burst, SCH one frame after FCCH).

Stream model: 51-multiframe on slot 0 -- FCCH at frames 0,10,20,30,40, SCH one
frame later, BCCH at frames 2..5, normal bursts elsewhere, idle frame 50 -- slots 1..7 carry
random normal bursts at equal power; per-stream sampling-clock error (resampling), carrier
error, AWGN, DC offset, uint8 quantisation.  RNG: numpy Philox keyed by (seed, dongle, arfcn).
"""
from __future__ import annotations

import math

import numpy as np

SYMBOL_RATE = (1625.0 / 6.0) * 1e3
OV = 8
FS = SYMBOL_RATE * OV
FRAME_OV = 1250 * OV  # 10 000 samples per TDMA frame at 8x
SLOT_OV = 1250        # 156.25 symbols * 8
BURST_BITS = 148

SCH_TRAINING_BITS = np.array(
    [1, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0,
     0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 0, 0,
     1, 0, 1, 0, 1, 1, 1, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0, 1, 1], dtype=np.int8)

# GSM 05.02 normal-burst training sequence code 0 (26 bits)
TSC0 = np.array([0, 0, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 1, 1], dtype=np.int8)

DEFAULT_SEED = 20260101


def _qfunc(x):
    return 0.5 * np.array([math.erfc(v / math.sqrt(2.0)) for v in np.ravel(x)]).reshape(np.shape(x))


def gmsk_frequency_pulse(bt=0.3, pulse_len=4, sps=OV):
    """Gaussian-filtered rectangular frequency pulse g[n], truncated to pulse_len symbols, sampled
    at sample centres, normalised so that sum(g) = 1/2 (one symbol moves the phase by pi/2)."""
    n = pulse_len * sps
    t = (np.arange(n) + 0.5) / sps - pulse_len / 2.0  # in symbol periods
    k = 2.0 * math.pi * bt / math.sqrt(math.log(2.0))
    g = _qfunc(k * (t - 0.5)) - _qfunc(k * (t + 0.5))
    return g * (0.5 / np.sum(g))


_G = gmsk_frequency_pulse()


def diff_precode(bits, prev=0):
    """d[i] = NOT(b[i] xor b[i-1]), b[-1] = prev (gsm_SCH_training_sequence_gen.m:32)."""
    b = np.concatenate([[prev], np.asarray(bits, dtype=np.int8)])
    return (1 - np.abs(np.diff(b))).astype(np.int8)


def gmsk_modulate(precoded_bits, sps=OV, g=None):
    """Unit-modulus GMSK of pre-coded bits (0 -> -1, 1 -> +1), phase starts at 0, output length
    len(bits)*sps (the pulse's own delay of pulse_len/2 symbols is inside the output, as with a
    streaming modulator)."""
    g = _G if g is None else g
    a = 2.0 * np.asarray(precoded_bits, dtype=np.float64) - 1.0
    up = np.zeros(len(a) * sps)
    up[::sps] = a
    f = np.convolve(up, g)[: len(up)]
    phase = math.pi * np.cumsum(f)
    return np.exp(1j * phase)


def sch_training_sequence(sps=OV):
    """512 x 1 complex template: stand-in for gsm_SCH_training_sequence_gen(8)."""
    return gmsk_modulate(diff_precode(SCH_TRAINING_BITS), sps)


def fir1(n, wn):
    """Own statement of fir1(n, Wn) (low-pass, Hamming window, unit DC gain), as used at
    gsm_sync_demod.m:34 / multi_rtl_sdr_gsm_FCCH_scanner.m:53."""
    k = np.arange(n + 1, dtype=np.float64)
    m = k - n / 2.0
    h = wn * np.sinc(wn * m)
    w = 0.54 - 0.46 * np.cos(2.0 * math.pi * k / n)
    h = h * w
    # MATLAB builds both the ideal response and the window as one half plus its mirror image, so the taps it
    # returns are symmetric to the last bit; do the same (the HIP front end has a faster kernel for such taps)
    half = (n + 1) // 2
    h[n + 1 - half:] = h[:half][::-1]
    return h / np.sum(h)


def _burst_bits(kind, rng):
    if kind == "F":
        return np.zeros(BURST_BITS, dtype=np.int8)
    if kind == "S":
        return np.concatenate([np.zeros(3, np.int8), rng.integers(0, 2, 39, dtype=np.int8),
                               SCH_TRAINING_BITS, rng.integers(0, 2, 39, dtype=np.int8),
                               np.zeros(3, np.int8)])
    # normal burst: 3 tail, 58 data(+stealing), 26 TSC, 58 data, 3 tail
    return np.concatenate([np.zeros(3, np.int8), rng.integers(0, 2, 58, dtype=np.int8), TSC0,
                           rng.integers(0, 2, 58, dtype=np.int8), np.zeros(3, np.int8)])


def _slot0_kind(frame_in_mf):
    f = frame_in_mf % 51
    if f == 50:
        return "I"
    if f % 10 == 0:
        return "F"
    if f % 10 == 1:
        return "S"
    return "N"


def _ramp(n_total, n_edge=8):
    w = np.ones(n_total)
    e = 0.5 - 0.5 * np.cos(math.pi * (np.arange(n_edge) + 0.5) / n_edge)
    w[:n_edge] = e
    w[-n_edge:] = e[::-1]
    return w


_RAMP = _ramp(BURST_BITS * OV)


def ideal_waveform(num_frames, start_frame, rng, bcch=True):
    """Complex baseband at exactly 8 samples/symbol, num_frames TDMA frames starting at frame
    `start_frame` of the 51-multiframe.  bcch=False gives a traffic-only carrier (no FCCH/SCH)."""
    n = num_frames * FRAME_OV
    x = np.zeros(n, dtype=np.complex128)
    # all bursts of the stream are modulated in one vectorised pass: bits -> (nb, 148)
    kinds = []
    for fr in range(num_frames):
        k0 = _slot0_kind(start_frame + fr) if bcch else "N"
        kinds.append(k0)
        kinds.extend(["N"] * 7)
    nb = len(kinds)
    # every burst is modulated with PRE guard bits in front and POST behind (zeros, like the tail
    # bits) so that the phase trajectory is already established at the burst's first sample; the
    # burst-local sample 0 is the instant bit 0's frequency pulse starts -- the same convention as
    # gmsk_modulate()/sch_training_sequence(), so the SCH training sequence sits 42*8 samples in.
    PRE, POST = 4, 4
    nbits = PRE + BURST_BITS + POST
    bits = np.zeros((nb, nbits), dtype=np.int8)
    for i, k in enumerate(kinds):
        bits[i, PRE:PRE + BURST_BITS] = _burst_bits("N" if k == "I" else k, rng)
    prev = np.concatenate([np.zeros((nb, 1), np.int8), bits[:, :-1]], axis=1)
    pre = 1 - np.abs(bits - prev)                       # diff_precode per burst
    a = 2.0 * pre.astype(np.float64) - 1.0
    up = np.zeros((nb, nbits * OV))
    up[:, ::OV] = a
    # causal convolution with g along time, truncated to the burst length
    f = np.zeros_like(up)
    for j, gj in enumerate(_G):
        f[:, j:] += gj * up[:, : up.shape[1] - j]
    ph = math.pi * np.cumsum(f, axis=1) + rng.uniform(0, 2 * math.pi, (nb, 1))
    b = np.exp(1j * ph[:, PRE * OV: (PRE + BURST_BITS) * OV]) * _RAMP[None, :]
    for i, k in enumerate(kinds):
        if k == "I":
            continue
        fr, sl = divmod(i, 8)
        s0 = fr * FRAME_OV + sl * SLOT_OV
        x[s0: s0 + BURST_BITS * OV] = b[i]
    return x


def make_stream(dongle=0, arfcn=0, num_frames=102, seed=DEFAULT_SEED, bcch=True,
                sampling_ppm=None, carrier_ppm=None, snr_db=None, carrier_freq=957.4e6,
                start_frame=None, frac_start=None, tx_key=None):
    """One capture: returns (raw uint8 interleaved I,Q of length 2*num_frames*10000, truth dict).
    tx_key: dongles given the same key receive the SAME transmission (payload, multiframe phase, start instant) through
    their own clocks, carrier offsets and noise -- what gsm_sync_demod.m's inter-dongle phase plot (:151-158) looks at."""
    rng = np.random.Generator(np.random.Philox(key=[int(seed), (int(dongle) << 20) ^ int(arfcn)]))
    eps_s = rng.uniform(-80.0, 80.0) if sampling_ppm is None else float(sampling_ppm)
    eps_c = rng.uniform(-40.0, 40.0) if carrier_ppm is None else float(carrier_ppm)
    snr = rng.uniform(15.0, 30.0) if snr_db is None else float(snr_db)
    sf = int(rng.integers(0, 51)) if start_frame is None else int(start_frame)
    fs0 = rng.uniform(0.0, FRAME_OV) if frac_start is None else float(frac_start)
    n = num_frames * FRAME_OV
    if tx_key is None:
        ideal = ideal_waveform(num_frames + 2, sf, rng, bcch=bcch)
    else:
        txr = np.random.Generator(np.random.Philox(key=[int(seed) ^ 0x5A5A5A5A, int(tx_key)]))
        if start_frame is None:
            sf = int(txr.integers(0, 51))
        if frac_start is None:
            fs0 = txr.uniform(0.0, FRAME_OV)
        ideal = ideal_waveform(num_frames + 2, sf, txr, bcch=bcch)
    # the dongle's clock runs (1+eps_s) fast => sample k is taken at ideal time k/(1+eps_s)... the
    # estimator's sign convention (FCCH_fine_correction.m:111-115): observed spacing = ideal*(1+e)
    # means the dongle takes MORE samples per GSM frame, i.e. query ideal time t_k = k/(1+e).
    e = eps_s * 1e-6
    t = fs0 + np.arange(n, dtype=np.float64) / (1.0 + e)
    i0 = np.floor(t).astype(np.int64)
    fr = t - i0
    # 4-point cubic (Catmull-Rom) interpolation of the 8x-oversampled waveform
    xm1, x0, x1, x2 = ideal[i0 - 1 + 1], ideal[i0 + 1], ideal[i0 + 2], ideal[i0 + 3]
    y = x0 + 0.5 * fr * (x1 - xm1 + fr * (2 * xm1 - 5 * x0 + 4 * x1 - x2 + fr * (3 * (x0 - x1) + x2 - xm1)))
    f_off = eps_c * 1e-6 * carrier_freq
    y = y * np.exp(1j * (2.0 * math.pi * f_off / FS * np.arange(n) + rng.uniform(0, 2 * math.pi)))
    amp = 70.0
    sigma = amp * 10.0 ** (-snr / 20.0) / math.sqrt(2.0)
    y = amp * y + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    dc = 127.4 + rng.uniform(-1.0, 1.0, 2)
    raw = np.empty(2 * n, dtype=np.float64)
    raw[0::2] = y.real + dc[0]
    raw[1::2] = y.imag + dc[1]
    raw = np.clip(np.floor(raw + 0.5), 0, 255).astype(np.uint8)
    truth = {"sampling_ppm": eps_s, "carrier_ppm": eps_c, "snr_db": snr, "start_frame": sf,
             "frac_start": fs0, "carrier_freq": carrier_freq, "bcch": bcch}
    return raw, truth


# ------------------------------------------------------------------------------------------------
# Host twin of the device-side capture expansion (csrc/kernels_frontend.h: k_synth_expand).
# ------------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def expand_capture(base, unit, seed=DEFAULT_SEED):
    """Capture `unit` of gsmcal_synth_expand_dev(base, ...): base is (K, 2N) uint8.  Bit-identical to the device kernel."""
    base = np.asarray(base, dtype=np.uint8)
    k, two_n = base.shape
    n = two_n // 2
    with np.errstate(over="ignore"):
        u = np.uint64(unit)
        sh = int(_mix64(np.uint64(seed) ^ (u * np.uint64(0xD1B54A32D192ED03))) % np.uint64(n))
        key = _mix64(np.uint64(seed) + np.uint64(0x632BE59BD9B4E019) * u)
    b = base[int(unit) % k].reshape(n, 2)
    src = (np.arange(n, dtype=np.int64) + sh) % n
    z = _mix64(key ^ np.arange(n, dtype=np.uint64))
    di = (z & np.uint64(3)).astype(np.int64)
    dq = ((z >> np.uint64(2)) & np.uint64(3)).astype(np.int64)
    out = np.empty((n, 2), dtype=np.int64)
    out[:, 0] = b[src, 0].astype(np.int64) + (di == 0) - (di == 1)
    out[:, 1] = b[src, 1].astype(np.int64) + (dq == 0) - (dq == 1)
    return np.clip(out, 0, 255).astype(np.uint8).reshape(-1)
