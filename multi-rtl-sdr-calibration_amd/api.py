"""Host-side mirror of the reference's MATLAB interface for the calibration path, over libgsmcal.so.

MATLAB is not available in the build image, so the host layer above the C ABI is Python.  Each
function keeps the reference function's name, argument order/meaning, 1-based positions and
sentinel returns (scalar -1.0 / inf, pos_info = [[-1, -1]]) so that tests read like calls of the
.m files:

    b = raw2iq(a)                                                   raw2iq.m:5
    r = chn_filter_8x_4x(s)                                         chn_filter_8x_4x.m:5
    r = chn_filter_4x(s)                                            chn_filter_4x.m:5
    [hit_flag,hit_idx,hit_avg_snr,hit_snr] = move_fft_snr_runtime_avg(s,mv_len,fft_len,th)
    [hit_flag,hit_idx,hit_snr] = specific_fft_snr_fix_avg(s,target_set,fft_len,th,avg_snr)
    [position,snr] = FCCH_coarse_position(s,decimation_ratio)
    [FCCH_pos,r,sampling_ppm,carrier_ppm] = FCCH_fine_correction(s,base_position,ov,carrier_freq)
    [pos_info,r,sampling_ppm] = SCH_corr_rate_correction(s,FCCH_pos,sch_training_sequence,ov)
    [r,carrier_ppm] = carrier_correct_post_SCH(s,pos_info,ov,carrier_freq)
    ppm_out = total_ppm_calculation(ppm_in)

Everything computes on the GPU through the C ABI; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import MAX_HITS, MAX_POS_ROWS, TABLE_COLS, GsmcalError

_CTX = {}


_VERBOSE = [os.environ.get("GSMCAL_VERBOSE") == "1"]


def set_verbose(on=True):
    """Print the reference's console diagnostics (FCCH_coarse_position.m:92-94, FCCH_fine_correction.m:66,116,156-161,190, ...:
    gsmcal_last_call_report) after every per-function call, as the .m files do.  Also switched on by GSMCAL_VERBOSE=1."""
    _VERBOSE[0] = bool(on)


def last_call_report(ctx=None):
    """The lines the .m file of the most recent per-function call would have disp()ed (a str, '\\n'-separated)."""
    ctx = ctx or default_context()
    n = ctx.lib.gsmcal_last_call_report(ctx.h, None, 0)
    if n <= 0:
        return ""
    buf = C.create_string_buffer(int(n) + 1)
    ctx.lib.gsmcal_last_call_report(ctx.h, buf, len(buf))
    return buf.value.decode()


def num2str(x):
    """MATLAB's num2str for a scalar / row vector, as the library formats its diagnostics (gsmcal_num2str)."""
    v = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel())
    lib = _lib.load()
    n = lib.gsmcal_num2str(_dp(v), len(v), None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.gsmcal_num2str(_dp(v), len(v), buf, len(buf))
    return buf.value.decode()


def _say(ctx):
    if _VERBOSE[0]:
        txt = last_call_report(ctx)
        if txt:
            print(txt, end="")


class Context:
    """One GPU + one HIP stream (gsmcal_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            rc = self.lib.gsmcal_ctx_create(int(device), C.byref(h))
        else:
            rc = self.lib.gsmcal_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc != 0:
            raise GsmcalError(f"gsmcal_ctx_create(device={device}) failed with {rc}: no usable gfx950 GPU "
                              f"(this package has no CPU fallback)")
        self.h = h
        self.device = device
        self.stream_handle = None if stream is None else int(stream)   # the caller's HIP stream (None: the context's own)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gsmcal_ctx_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    _E_NAMES = {-1: "GSMCAL_E_ARG (an argument is null, zero or out of its documented range: include/gsmcal.h)", -2: "GSMCAL_E_HIP",
                -3: "GSMCAL_E_NO_DEVICE", -4: "GSMCAL_E_CAPACITY (an output buffer's capacity argument is too small)",
                -5: "GSMCAL_E_INDEX (MATLAB would raise 'index exceeds matrix dimensions')", -6: "GSMCAL_E_UNSUPPORTED"}

    def check(self, rc, what):
        if rc < 0:
            msg = self.lib.gsmcal_last_error(self.h)
            text = msg.decode() if msg else ""
            raise GsmcalError(f"{what} failed with {rc} = {self._E_NAMES.get(rc, '?')}" + (f": {text}" if text else ""))
        return rc

    def sync(self):
        self.check(self.lib.gsmcal_sync(self.h), "gsmcal_sync")

    def set_pipeline_depth(self, depth):
        """gsmcal_ctx_set_pipeline_depth: up to `depth` consecutive single-lane calibrate_batch_dev calls in flight (the front
        end of call i+1 under the tail of call i); outputs of call i complete at call i+depth or sync().  1 = off."""
        self.check(self.lib.gsmcal_ctx_set_pipeline_depth(self.h, int(depth)), "gsmcal_ctx_set_pipeline_depth")

    def pipeline_depth(self):
        return int(self.lib.gsmcal_ctx_get_pipeline_depth(self.h))

    def pipeline_queues(self):
        """gsmcal_ctx_pipeline_queues: internal streams seen running side by side (0 before the first call in flight)"""
        return int(self.lib.gsmcal_ctx_pipeline_queues(self.h))

    def fused_tail_reruns(self):
        """how often gsmcal_sync / a host-buffer call re-ran calls with the four-launch tail after a fused tail timed out"""
        return int(self.lib.gsmcal_fused_tail_reruns(self.h))

    def fused_tail_stats(self):
        """(batch calls that took the fused tail, calls the one-fused-tail-per-device gate sent to the four-launch tail)"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self.check(self.lib.gsmcal_fused_tail_stats(self.h, C.byref(a), C.byref(b)), "gsmcal_fused_tail_stats")
        return int(a.value), int(b.value)

    # ---- thresholds (gsmcal_params: the constants the reference hard-codes) ----
    def get_params(self):
        p = _lib.Params()
        self.check(self.lib.gsmcal_get_params(self.h, C.byref(p)), "gsmcal_get_params")
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(f"gsmcal_params has no field {k}")
            setattr(p, k, v)
        self.check(self.lib.gsmcal_set_params(self.h, C.byref(p)), "gsmcal_set_params")

    # ---- device memory ----
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.gsmcal_dev_alloc(self.h, int(nbytes), C.byref(p)), "gsmcal_dev_alloc")
        return p.value

    def free(self, ptr):
        self.check(self.lib.gsmcal_dev_free(self.h, C.c_void_p(ptr)), "gsmcal_dev_free")

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.gsmcal_memcpy_h2d(self.h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes),
                   "gsmcal_memcpy_h2d")

    def d2h(self, arr, dptr):
        assert arr.flags.c_contiguous
        self.check(self.lib.gsmcal_memcpy_d2h(self.h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), arr.nbytes),
                   "gsmcal_memcpy_d2h")

    # ---- profiling ----
    def profile_enable(self, on=True):
        self.check(self.lib.gsmcal_profile_enable(self.h, 1 if on else 0), "gsmcal_profile_enable")

    def profile_filter(self, substr=None):
        self.check(self.lib.gsmcal_profile_filter(self.h, substr.encode() if substr else None), "gsmcal_profile_filter")

    def profile_reset(self):
        self.check(self.lib.gsmcal_profile_reset(self.h), "gsmcal_profile_reset")

    def profile_get(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n = (C.c_long * cap)()
        k = self.check(self.lib.gsmcal_profile_get(self.h, cap, names, ms, n), "gsmcal_profile_get")
        return {names[i].decode(): (ms[i], n[i]) for i in range(min(k, cap))}


def default_context(device=0):
    if device not in _CTX:
        _CTX[device] = Context(device)
    return _CTX[device]


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


def _cplx_in(s):
    """complex (n,) or (n,d) -> contiguous column-major interleaved doubles, returns (buf, n, d)."""
    s = np.asarray(s)
    if s.ndim == 1:
        s = s[:, None]
    n, d = s.shape
    buf = np.ascontiguousarray(s.T.astype(np.complex128))  # (d, n): each column contiguous
    return buf, n, d


# ------------------------------------------------------------------------------------------------
def raw2iq(a, ctx=None):
    """b = raw2iq(a) -- raw2iq.m:5-8.  a: (2N,) or (2N,D) byte values (uint8 or doubles)."""
    ctx = ctx or default_context()
    a = np.asarray(a)
    squeeze = a.ndim == 1
    if squeeze:
        a = a[:, None]
    rows, d = a.shape
    out = np.empty((d, rows // 2), dtype=np.complex128)
    if a.dtype == np.uint8:
        buf = np.ascontiguousarray(a.T)
        rc = ctx.lib.gsmcal_raw2iq_u8(ctx.h, buf.ctypes.data_as(_lib.c_u8_p), rows, d, _dp(out))
    else:
        buf = np.ascontiguousarray(a.T.astype(np.float64))
        rc = ctx.lib.gsmcal_raw2iq(ctx.h, _dp(buf), rows, d, _dp(out))
    ctx.check(rc, "raw2iq")
    return out[0] if squeeze else out.T


def filter(coef, s, decim=1, ctx=None):  # noqa: A001 - mirrors MATLAB's filter(coef,1,s)
    """r = filter(coef,1,s) column-wise (gsm_sync_demod.m:110), optionally r(1:decim:end,:)."""
    ctx = ctx or default_context()
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    squeeze = np.asarray(s).ndim == 1
    buf, n, d = _cplx_in(s)
    nd = (n + decim - 1) // decim
    out = np.empty((d, nd), dtype=np.complex128)
    ctx.check(ctx.lib.gsmcal_filter(ctx.h, _dp(coef), len(coef), _dp(buf), n, d, decim, _dp(out)), "filter")
    return out[0] if squeeze else out.T


def chn_filter_8x_4x(s, num=None, ctx=None):
    """r = chn_filter_8x_4x(s) -- chn_filter_8x_4x.m:5-15 (60 built-in taps unless `num` is given)."""
    ctx = ctx or default_context()
    squeeze = np.asarray(s).ndim == 1
    buf, n, d = _cplx_in(s)
    out = np.empty((d, (n + 1) // 2), dtype=np.complex128)
    if num is None:
        rc = ctx.lib.gsmcal_chn_filter_8x_4x(ctx.h, _dp(buf), n, d, None, 0, _dp(out))
    else:
        num = np.ascontiguousarray(num, dtype=np.float64)
        rc = ctx.lib.gsmcal_chn_filter_8x_4x(ctx.h, _dp(buf), n, d, _dp(num), len(num), _dp(out))
    ctx.check(rc, "chn_filter_8x_4x")
    return out[0] if squeeze else out.T


def chn_filter_4x(s, num=None, ctx=None):
    """r = chn_filter_4x(s) -- chn_filter_4x.m:5-13 (30 built-in taps of gsm_chn_filter_4x.fda unless `num` is given)."""
    ctx = ctx or default_context()
    squeeze = np.asarray(s).ndim == 1
    buf, n, d = _cplx_in(s)
    out = np.empty((d, n), dtype=np.complex128)
    if num is None:
        rc = ctx.lib.gsmcal_chn_filter_4x(ctx.h, _dp(buf), n, d, None, 0, _dp(out))
    else:
        num = np.ascontiguousarray(num, dtype=np.float64)
        rc = ctx.lib.gsmcal_chn_filter_4x(ctx.h, _dp(buf), n, d, _dp(num), len(num), _dp(out))
    ctx.check(rc, "chn_filter_4x")
    return out[0] if squeeze else out.T


def move_fft_snr_runtime_avg(s, mv_len, fft_len, th, ctx=None):
    ctx = ctx or default_context()
    buf, n, _ = _cplx_in(np.asarray(s).ravel())
    hf = C.c_int()
    hi, ha, hs = C.c_double(), C.c_double(), C.c_double()
    ctx.check(ctx.lib.gsmcal_move_fft_snr_runtime_avg(ctx.h, _dp(buf), n, int(mv_len), int(fft_len), float(th),
                                                      C.byref(hf), C.byref(hi), C.byref(ha), C.byref(hs)),
              "move_fft_snr_runtime_avg")
    return bool(hf.value), (int(hi.value) if hf.value else -1), ha.value, hs.value


def specific_fft_snr_fix_avg(s, target_set, fft_len, th, avg_snr, ctx=None):
    ctx = ctx or default_context()
    buf, n, _ = _cplx_in(np.asarray(s).ravel())
    ts = (C.c_double * 2)(float(target_set[0]), float(target_set[1]))
    hf = C.c_int()
    hi, hs = C.c_double(), C.c_double()
    ctx.check(ctx.lib.gsmcal_specific_fft_snr_fix_avg(ctx.h, _dp(buf), n, ts, int(fft_len), float(th),
                                                      float(avg_snr), C.byref(hf), C.byref(hi), C.byref(hs)),
              "specific_fft_snr_fix_avg")
    return bool(hf.value), (int(hi.value) if hf.value else -1), hs.value


def FCCH_coarse_position(s, decimation_ratio, ctx=None):
    """[position, snr] = FCCH_coarse_position(s, decimation_ratio); (-1.0, -1.0) when nothing found."""
    ctx = ctx or default_context()
    buf, n, _ = _cplx_in(np.asarray(s).ravel())
    pos = np.zeros(MAX_HITS)
    snr = np.zeros(MAX_HITS)
    cnt = C.c_int()
    rc = ctx.check(ctx.lib.gsmcal_FCCH_coarse_position(ctx.h, _dp(buf), n, int(decimation_ratio), _dp(pos),
                                                       _dp(snr), MAX_HITS, C.byref(cnt)), "FCCH_coarse_position")
    _say(ctx)
    if rc == 1:
        return -1.0, -1.0
    return pos[:cnt.value].copy(), snr[:cnt.value].copy()


def _r_in(s):
    """A stream argument that may be the reference's r = -1 sentinel."""
    if np.ndim(s) == 0:
        return None, 0
    buf, n, _ = _cplx_in(np.asarray(s).ravel())
    return buf, n


def FCCH_fine_correction(s, base_position, oversampling_ratio, carrier_freq, ctx=None, want_r=True):
    """[FCCH_pos, r, sampling_ppm, carrier_ppm] = FCCH_fine_correction(...) -- FCCH_fine_correction.m:5."""
    ctx = ctx or default_context()
    buf, n = _r_in(s)
    bp = np.ascontiguousarray(np.atleast_1d(np.asarray(base_position, dtype=np.float64)))
    pos = np.zeros(MAX_HITS)
    npos = C.c_int()
    r = np.empty(n if want_r else 0, dtype=np.complex128)
    lr = C.c_long()
    sp, cp = C.c_double(), C.c_double()
    ctx.check(ctx.lib.gsmcal_FCCH_fine_correction(ctx.h, _dp(buf), n, _dp(bp), len(bp), int(oversampling_ratio),
                                                  float(carrier_freq), _dp(pos), MAX_HITS, C.byref(npos),
                                                  _dp(r) if want_r else None, len(r), C.byref(lr),
                                                  C.byref(sp), C.byref(cp)), "FCCH_fine_correction")
    _say(ctx)
    fpos = pos[:npos.value].copy()
    if npos.value == 1 and fpos[0] == -1.0:
        fpos = -1.0
    rr = -1.0 if lr.value < 0 else (r[:lr.value] if want_r else lr.value)
    return fpos, rr, sp.value, cp.value


def SCH_corr_rate_correction(s, FCCH_pos, sch_training_sequence, oversampling_ratio, ctx=None, want_r=True):
    """[pos_info, r, sampling_ppm] = SCH_corr_rate_correction(...) -- SCH_corr_rate_correction.m:5."""
    ctx = ctx or default_context()
    buf, n = _r_in(s)
    fp = np.ascontiguousarray(np.atleast_1d(np.asarray(FCCH_pos, dtype=np.float64)))
    ts = np.ascontiguousarray(np.asarray(sch_training_sequence, dtype=np.complex128).ravel())
    pi = np.zeros((2, MAX_POS_ROWS))
    nrows = C.c_int()
    r = np.empty(n if want_r else 0, dtype=np.complex128)
    lr = C.c_long()
    sp = C.c_double()
    ctx.check(ctx.lib.gsmcal_SCH_corr_rate_correction(ctx.h, _dp(buf) if buf is not None else None, n, _dp(fp),
                                                      len(fp), _dp(ts), len(ts), int(oversampling_ratio), _dp(pi),
                                                      MAX_POS_ROWS, C.byref(nrows), _dp(r) if want_r else None,
                                                      len(r), C.byref(lr), C.byref(sp)), "SCH_corr_rate_correction")
    _say(ctx)
    pos_info = pi[:, :nrows.value].T.copy()
    rr = -1.0 if lr.value < 0 else (r[:lr.value] if want_r else lr.value)
    return pos_info, rr, sp.value


def carrier_correct_post_SCH(s, pos_info, oversampling_ratio, carrier_freq, ctx=None, want_r=True):
    """[r, carrier_ppm] = carrier_correct_post_SCH(...) -- carrier_correct_post_SCH.m:5."""
    ctx = ctx or default_context()
    buf, n = _r_in(s)
    pi = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    rows = pi.shape[0]
    pic = np.ascontiguousarray(pi.T)  # column-major, ld = rows
    r = np.empty(n if want_r else 0, dtype=np.complex128)
    lr = C.c_long()
    cp = C.c_double()
    ctx.check(ctx.lib.gsmcal_carrier_correct_post_SCH(ctx.h, _dp(buf) if buf is not None else None, n, _dp(pic),
                                                      rows, rows, int(oversampling_ratio), float(carrier_freq),
                                                      _dp(r) if want_r else None, len(r), C.byref(lr), C.byref(cp)),
              "carrier_correct_post_SCH")
    _say(ctx)
    rr = -1.0 if lr.value < 0 else (r[:lr.value] if want_r else lr.value)
    return rr, cp.value


def SCH_equalise(s, pos_info, training_sequence, oversampling_ratio, ctx=None):
    """Front end of SCH_demod(s, pos_info, training_sequence, ov) -- SCH_demod.m:53-59,79-90: the equalised burst of
    every SCH row of pos_info, shape (num_sch, 194*ov).  pos_info all -1 (:8-11) -> None."""
    ctx = ctx or default_context()
    buf, n = _r_in(s)
    pi = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    rows = pi.shape[0]
    pic = np.ascontiguousarray(pi.T)
    ts = np.ascontiguousarray(np.asarray(training_sequence, dtype=np.complex128).ravel())
    L = 194 * int(oversampling_ratio)
    cap = int(np.sum(pi[:, 1] == 1)) if pi.shape[1] > 1 else 0
    out = np.empty((max(cap, 1), L), dtype=np.complex128)
    nb, lf = C.c_int(), C.c_int()
    rc = ctx.check(ctx.lib.gsmcal_SCH_equalise(ctx.h, _dp(buf) if buf is not None else None, n, _dp(pic), rows, rows, _dp(ts),
                                               len(ts), int(oversampling_ratio), _dp(out), cap, C.byref(nb), C.byref(lf)),
                   "SCH_equalise")
    if rc == 10:
        return None
    return out[:nb.value]


def total_ppm_calculation(ppm_in):
    lib = _lib.load()
    p = np.ascontiguousarray(np.atleast_1d(np.asarray(ppm_in, dtype=np.float64)))
    out = C.c_double()
    rc = lib.gsmcal_total_ppm_calculation(_dp(p), len(p), C.byref(out))
    if rc < 0:
        raise GsmcalError(f"total_ppm_calculation failed with {rc}")
    if rc == 12 and _VERBOSE[0]:
        print("total PPM calculation: No valid PPM input!")          # total_ppm_calculation.m:8
    return out.value


# ------------------------------------------------------------------------------------------------
# batched hot path
# ------------------------------------------------------------------------------------------------
TABLE_FIELDS = ("sampling_ppm_fcch", "sampling_ppm_sch", "carrier_ppm_fcch", "carrier_ppm_post",
                "total_sampling_ppm", "total_carrier_ppm", "n_fcch", "n_pos_rows", "first_fcch_pos", "status")


def frontend_batch(raw, coef, decim, ctx=None):
    """raw: (D, 2N) uint8 -> (D, ceil(N/decim)) complex: raw2iq + filter + r(1:decim:end)."""
    ctx = ctx or default_context()
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    d, two_n = raw.shape
    n = two_n // 2
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    out = np.empty((d, (n + decim - 1) // decim), dtype=np.complex128)
    ctx.check(ctx.lib.gsmcal_frontend_batch(ctx.h, raw.ctypes.data_as(_lib.c_u8_p), d, n, _dp(coef), len(coef),
                                            int(decim), _dp(out)), "frontend_batch")
    return out


def fcch_scan_batch(raw, coef, ctx=None):
    """Scanner detect loop for D captures -> dict(snr, num_hit, positions, pos_snr, counts)."""
    ctx = ctx or default_context()
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    d, two_n = raw.shape
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    snr = np.zeros(d)
    nh = np.zeros(d)
    pos = np.zeros((d, MAX_HITS))
    psnr = np.zeros((d, MAX_HITS))
    cnt = np.zeros(d, dtype=np.int32)
    ctx.check(ctx.lib.gsmcal_fcch_scan_batch(ctx.h, raw.ctypes.data_as(_lib.c_u8_p), d, two_n // 2, _dp(coef),
                                             len(coef), _dp(snr), _dp(nh), _dp(pos), _dp(psnr),
                                             cnt.ctypes.data_as(_lib.c_int_p)), "fcch_scan_batch")
    return {"snr": snr, "num_hit": nh, "positions": pos, "pos_snr": psnr, "counts": cnt}


def calibrate_batch(raw, coef, sch_training_sequence, carrier_freq, want_r=False, ctx=None):
    """gsm_sync_demod.m:107-124 for D streams: raw (D, 2N) uint8 -> dict(table, pos_info, r_correct, r_len)."""
    ctx = ctx or default_context()
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    d, two_n = raw.shape
    n = two_n // 2
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    ts = np.ascontiguousarray(np.asarray(sch_training_sequence, dtype=np.complex128).ravel())
    cf = np.ascontiguousarray(np.broadcast_to(np.asarray(carrier_freq, dtype=np.float64), (d,)))
    table = np.zeros((d, TABLE_COLS))
    pos_info = np.zeros((d, 2, MAX_POS_ROWS))
    r_len = np.zeros(d, dtype=np.int64)
    r = np.empty((d, n), dtype=np.complex128) if want_r else None
    ctx.check(ctx.lib.gsmcal_calibrate_batch(ctx.h, raw.ctypes.data_as(_lib.c_u8_p), d, n, _dp(coef), len(coef),
                                             _dp(ts), len(ts), _dp(cf), _dp(table), _dp(pos_info),
                                             _dp(r) if want_r else None, r_len.ctypes.data_as(_lib.c_long_p)),
              "calibrate_batch")
    out = {"table": table, "pos_info_raw": pos_info, "r_len": r_len, "r_correct": r}
    rows = []
    for i in range(d):
        k = int(table[i, 7])
        if table[i, 8] == -1.0:
            # the reference's all -1 sentinel, in the shape of the exit taken: [-1 -1] (SCH_corr_rate_correction.m:9,:61) or
            # the -ones(3*num_fcch_hit,2) pre-allocation of :32 returned by the :84 / :106-112 exits
            rows.append(-np.ones((k, 2)))
        else:
            rows.append(pos_info[i, :, :k].T.copy())
    out["pos_info"] = rows
    return out


def calibrate_batch_dev(d_raw, d, n, coef, sch_training_sequence, carrier_freq, d_table, d_pos_info=None, d_r_correct=None,
                        d_r_len=None, ctx=None):
    """Device-pointer form of calibrate_batch: only enqueues on the context's stream (ctx.sync() before reading the outputs).
    d_raw: [d][2n] bytes; d_table: [d][TABLE_COLS] doubles in any memory the GPU can store to (device or pinned host);
    optional d_pos_info [d][2][MAX_POS_ROWS], d_r_correct [d][n] complex, d_r_len [d] int64.  carrier_freq: scalar or [d]."""
    ctx = ctx or default_context()
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    ts = np.ascontiguousarray(np.asarray(sch_training_sequence, dtype=np.complex128).ravel())
    cf = np.ascontiguousarray(np.broadcast_to(np.asarray(carrier_freq, dtype=np.float64), (int(d),)))
    vp = lambda p: C.c_void_p(p) if p else None  # noqa: E731
    ctx.check(ctx.lib.gsmcal_calibrate_batch_dev(ctx.h, C.c_void_p(d_raw), int(d), int(n), _dp(coef), len(coef),
                                                 _dp(ts), len(ts), _dp(cf),
                                                 C.c_void_p(d_table), vp(d_pos_info), vp(d_r_correct), vp(d_r_len)),
              "calibrate_batch_dev")


def fcch_scan_batch_dev(d_raw, d, n, coef, d_snr_numhit, d_positions=None, d_pos_snr=None, d_counts=None, ctx=None):
    """Device-pointer form of fcch_scan_batch: only enqueues (call ctx.sync() before reading the outputs).
    d_raw: [d][2n] bytes; d_snr_numhit: [d][2] doubles; optional [d][MAX_HITS] doubles x2 and [d] ints."""
    ctx = ctx or default_context()
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    vp = lambda p: C.c_void_p(p) if p else None  # noqa: E731
    ctx.check(ctx.lib.gsmcal_fcch_scan_batch_dev(ctx.h, C.c_void_p(d_raw), int(d), int(n), _dp(coef), len(coef),
                                                 C.c_void_p(d_snr_numhit), vp(d_positions), vp(d_pos_snr), vp(d_counts)),
              "fcch_scan_batch_dev")


def synth_expand_dev(d_base, k, n, d_out, d, first_unit=0, seed=20260101, ctx=None):
    """Synthetic-input utility: expand k base captures on the device into d distinct ones (see include/gsmcal.h;
    synth.expand_capture is the host twin)."""
    ctx = ctx or default_context()
    ctx.check(ctx.lib.gsmcal_synth_expand_dev(ctx.h, C.c_void_p(d_base), int(k), int(n), C.c_void_p(d_out), int(d),
                                              int(first_unit), int(seed)), "synth_expand_dev")


def last_batch_snr(stream, ctx=None):
    """Parity tap: (snr table of `stream` from the last batch call, number of moving-search windows at its head)."""
    ctx = ctx or default_context()
    buf = np.zeros(1 << 16)
    n_tab, n_mov = C.c_long(0), C.c_long(0)
    ctx.check(ctx.lib.gsmcal_last_batch_snr(ctx.h, int(stream), _dp(buf), len(buf), C.byref(n_tab), C.byref(n_mov)),
              "last_batch_snr")
    return buf[: min(n_tab.value, len(buf))].copy(), int(n_mov.value)


def last_batch_details(d, ctx=None):
    """Intermediates of the last batch call (coarse/fine/SCH positions per stream) for parity tests."""
    ctx = ctx or default_context()
    arrs = [np.zeros((d, MAX_HITS)) for _ in range(5)]
    counts = np.zeros((d, 5), dtype=np.int32)
    ctx.check(ctx.lib.gsmcal_last_batch_details(ctx.h, d, *[_dp(a) for a in arrs],
                                                counts.ctypes.data_as(_lib.c_int_p)), "last_batch_details")
    names = ("coarse_pos", "coarse_snr", "fine_first", "fcch_pos", "sch_first")
    out = {k: a for k, a in zip(names, arrs)}
    out["counts"] = counts
    return out
