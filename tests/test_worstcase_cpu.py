"""CPU checks of the constructed worst cases (tests/worstcase.py): every construction must put the ORACLE's decision where it
was designed to be, by the designed margin -- otherwise the GPU tests that replay them (tests/test_gpu_worstcase.py) would
prove nothing."""
import numpy as np
import pytest

import worstcase as wc
from oracle import gsmcal_oracle as o

FC = 957.4e6


@pytest.mark.parametrize("eps", [1e-6, -1e-9, 1e-12, -1e-12])
def test_two_tone_windows_decide_by_the_designed_ratio(eps):
    geoms = wc.TWO_TONE_GEOMETRIES["interior"][:3] + wc.TWO_TONE_GEOMETRIES["edge"][:3]
    s, base, want = wc.two_tone_stream([g + (eps,) for g in geoms])
    info = {}
    o.FCCH_fine_correction(s, base, 8, FC, info=info)
    assert np.array_equal(info["first_round_pos"], want)
    for i, (tA, kA, tB, kB) in enumerate(geoms):
        pk, kk = wc.fine_peak_map(s, int(base[i]))
        top = np.argsort(pk)[::-1]
        win, lose = (tB, tA) if eps > 0 else (tA, tB)
        assert top[0] == win and top[1] == lose and {int(kk[tA]), int(kk[tB])} == {kA, kB}
        ratio = pk[lose] / pk[win] - 1.0
        assert -2.5 * abs(eps) < ratio < -0.4 * abs(eps), (geoms[i], ratio)      # the loser sits |eps| below, to within the FFT's own rounding
        assert pk[top[2]] / pk[win] - 1.0 < -5e-6                                # nothing else comes close


@pytest.mark.parametrize("where", ["first", "hop"])
@pytest.mark.parametrize("delta", [1e-7, -1e-7, 1e-10, -1e-10])
def test_coarse_threshold_streams_sit_on_the_threshold(where, delta):
    s, info = wc.coarse_threshold_stream(delta, where)
    assert np.sign(info["margin"]) == np.sign(delta) and 0.5 * abs(delta) <= abs(info["margin"]) <= 4.0 * abs(delta)
    pos, snr = o.FCCH_coarse_position(s, 8)
    flip, _ = wc.coarse_threshold_stream(-delta, where)
    pos2, _ = o.FCCH_coarse_position(flip, 8)
    assert len(pos) >= 3 and len(pos2) >= 3
    assert not np.array_equal(pos[:2], pos2[:2]), "the decision at the threshold must change the outcome"


def test_degenerate_captures_are_what_they_say():
    caps = wc.degenerate_captures(24)
    assert set(np.unique(caps["clipped_noise"])) == {0, 255} and set(np.unique(caps["rail_to_rail"])) == {0, 255}
    r = o.raw2iq(caps["constant_128"].astype(np.float64))
    assert np.all(r == 0)                                                       # exact zeros: every window SNR is 0/0
    with np.errstate(all="ignore"):
        assert np.all(np.isnan(wc.snr_series(r[:64])))
