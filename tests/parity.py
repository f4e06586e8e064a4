"""Shared helpers for the parity tests: compare the HIP path with the CPU oracle on one stream."""
import math

import numpy as np

PPM_RTOL = 1e-6   # BASELINE.json north_star: ppm within 1e-6 relative
PPM_ATOL = 1e-9   # ppm, SURVEY 8d: absolute floor where the oracle value is ~0
SNR_ATOL = 1e-8   # dB: SNRs are reported values, not decisions


def ppm_close(a, b):
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= PPM_RTOL * abs(b) + PPM_ATOL


def assert_ppm(a, b, what):
    assert ppm_close(float(a), float(b)), f"{what}: gpu {a!r} vs oracle {b!r}"


def assert_positions(a, b, what):
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    assert a.shape == b.shape and np.array_equal(a, b), f"{what}: gpu {a} vs oracle {b}"


def compare_stream(orc, row, det, i, pos_info):
    """orc: oracle.calibrate_stream() dict; row: table row; det: last_batch_details; i: stream idx."""
    cnt = det["counts"][i]
    # coarse
    n = cnt[0]
    if np.ndim(orc["coarse_pos"]) and orc["coarse_pos"][0] != -1.0:
        assert_positions(det["coarse_pos"][i, :n], orc["coarse_pos"], "coarse_pos")
        assert np.allclose(det["coarse_snr"][i, :n], orc["coarse_snr"], rtol=0, atol=SNR_ATOL), "coarse_snr"
    else:
        assert n == 0, "coarse: oracle found nothing, gpu did"
    assert_positions(det["fine_first"][i, :cnt[1]], orc["fine_first_round_pos"], "fine first-round FCCH_pos")
    if orc["fcch_pos"].shape == (1,) and orc["fcch_pos"][0] == -1.0:
        assert cnt[2] == -1, "FCCH_pos sentinel"
    else:
        assert_positions(det["fcch_pos"][i, :cnt[2]], orc["fcch_pos"], "FCCH_pos")
    if orc.get("sch_edge_abort"):
        # the reference stops at the first edge peak (SCH_corr_rate_correction.m:59-63); the HIP path evaluates every
        # window in parallel: the positions up to and including the offending one must agree
        k = len(orc["sch_first_round_pos"])
        assert_positions(det["sch_first"][i, :k], orc["sch_first_round_pos"], "SCH first-round positions (prefix)")
    else:
        assert_positions(det["sch_first"][i, :cnt[3]], orc["sch_first_round_pos"], "SCH first-round positions")
    opi = orc["pos_info"]
    if np.all(opi == -1):
        # the reference's sentinel keeps the shape of the exit taken: [-1 -1] or -ones(3*num_fcch_hit, 2)
        assert pos_info.shape == opi.shape and np.all(pos_info == -1), f"pos_info sentinel shape {pos_info.shape} vs {opi.shape}"
    else:
        assert_positions(pos_info, opi, "pos_info")
    assert_ppm(row[0], orc["sampling_ppm"][0], "sampling_ppm(1)")
    assert_ppm(row[1], orc["sampling_ppm"][1], "sampling_ppm(2)")
    assert_ppm(row[2], orc["carrier_ppm"][0], "carrier_ppm(1)")
    assert_ppm(row[3], orc["carrier_ppm"][1], "carrier_ppm(2)")
    assert_ppm(row[4], orc["total_sampling_ppm"], "total sampling ppm")
    assert_ppm(row[5], orc["total_carrier_ppm"], "total carrier ppm")


# ---- helpers for tests that need many streams: worker functions for a SPAWNED process pool (a pytest process that has
# already initialised HIP must not fork) --------------------------------------------------------------------------------
def gen_stream_job(job):
    """job = (dongle, num_frames, kwargs) -> uint8 capture"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    dongle, frames, kw = job
    from gsmcal import synth
    return synth.make_stream(dongle=dongle, num_frames=frames, **kw)[0]


def oracle_job(job):
    """job = (raw, coef, ts, fc) -> oracle.calibrate_stream dict (without the corrected stream)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import gsmcal_oracle as oracle
    raw, coef, ts, fc = job
    return oracle.calibrate_stream(raw, coef, ts, fc)


def pool_map(fn, jobs, max_workers=32):
    """fn over jobs on spawned worker processes (safe after the parent initialised the GPU); serial when only one core"""
    import multiprocessing as mp
    import os
    from concurrent.futures import ProcessPoolExecutor
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(ncpu, max_workers, len(jobs)))
    if workers == 1:
        return [fn(j) for j in jobs]
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(fn, jobs, chunksize=1))


def oracle_job_safe(job):
    """oracle_job that reports the reference's index errors (MATLAB would raise: the ABI answers GSMCAL_E_INDEX) instead of
    raising in the worker: -> (dict or None, message or None)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import gsmcal_oracle as oracle
    raw, coef, ts, fc = job
    try:
        return oracle.calibrate_stream(raw, coef, ts, fc), None
    except oracle.MatlabIndexError as e:
        return None, str(e)


def scan_units_job(job):
    """job = (base captures, first unit, count, coef): oracle.scan_capture over the host twins of the device-expanded captures
    [first, first + count) -> list of dicts (snr, num_hit, coarse_pos, coarse_snr)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from gsmcal import synth
    from oracle import gsmcal_oracle as oracle
    base, first, count, coef = job
    out = []
    for u in range(first, first + count):
        r = oracle.scan_capture(synth.expand_capture(base, u), coef)
        out.append({k: r[k] for k in ("snr", "num_hit", "coarse_pos", "coarse_snr")})
    return out
