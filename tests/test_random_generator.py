"""CPU test of the GPU suite's random-system generator (tests/test_gpu_random_parity.py): the three properties the closing campaign of
round 6 showed it needs (profiles/r06_v5_closing_campaign.txt) -- no bare pair force beyond the fixed-point accumulators' range, no pair
straddling the cutoff between an f32 and an f64 evaluation of d^2 -- on the very seeds that were flagged.  No GPU, no product code."""
import numpy as np
import pytest

import test_gpu_random_parity as T


@pytest.mark.parametrize("seed", [20148, 21158, 4])
def test_bare_pair_forces_stay_inside_the_fixed_point_range(seed):
    s = T.draw_system(seed, gentle=False)
    g_lj, g_es = T.pair_force_matrix(s["x"], s["params"], s["box"], s["cutoff"], s["beta"])
    assert (g_lj + g_es).max() <= 2.0 ** 25
    assert np.all(s["params"][:, 1] > 0)


def test_an_f32_cases_cutoff_moves_off_straddling_pairs():
    s = T.draw_system(21355, gentle=True)
    x = s["x"].astype(np.float32).astype(np.float64)
    prm = s["params"].astype(np.float32).astype(np.float64)
    c = T.cutoff_off_straddling_pairs(x, prm, s["box"], s["cutoff"])
    assert c > s["cutoff"] and c - s["cutoff"] < 2e-3  # (pair (7, 381) of this seed: d^2 - cutoff^2 = -8.8e-8)
    L = np.diagonal(s["box"])
    d = x[:, None, :] - x[None, :, :]
    d -= L * np.rint(d / L)
    d2 = (d ** 2).sum(-1) + (prm[:, 3][:, None] - prm[None, :, 3]) ** 2
    for dt in (np.float32, np.float64):  # the same pairs inside in either precision
        inside64 = d2 < c * c
        inside32 = d2.astype(np.float32) < np.float32(c) * np.float32(c)
        assert np.array_equal(inside64, inside32)
    s2 = T.draw_system(3, gentle=True)  # a seed without such a pair keeps its cutoff
    assert T.cutoff_off_straddling_pairs(s2["x"].astype(np.float32).astype(np.float64), s2["params"].astype(np.float32).astype(np.float64), s2["box"], s2["cutoff"]) == s2["cutoff"]
