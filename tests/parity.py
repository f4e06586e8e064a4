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
