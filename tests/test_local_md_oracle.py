"""CPU checks of oracle/local_md.py: the pieces of local MD's atom selection that have published or reference-side answers.

 * MT19937: the C++ standard's known answer (the 10000th output of a default-constructed std::mt19937 is 4123659995,
   [rand.predef]) and numpy's RandomState (same generator, same integer seeding).
 * uniform_int_distribution (libstdc++ 11, Lemire): range, determinism, near-uniform counts, and the rejection branch.
 * acceptance probabilities against the flat-bottom energy the reference's own test recomputes
   (tests/test_md.py:713-790 uses exp(-U_flat_bottom / kT) with U from potentials/bonded.py:219-253)."""
import numpy as np

from oracle import local_md as olm


def test_mt19937_known_answer_and_numpy():
    g = olm.MT19937(5489)
    for _ in range(9999):
        g()
    assert g() == 4123659995
    for seed in (0, 1, 2022, 2**32 - 1):
        g = olm.MT19937(seed)
        want = np.frombuffer(np.random.RandomState(seed).bytes(4 * 700), dtype="<u4")
        got = np.array([g() for _ in range(700)], dtype=np.uint32)
        np.testing.assert_array_equal(got, want)


def test_uniform_int_below():
    class Fixed:
        def __init__(self, values):
            self.values = list(values)

        def __call__(self):
            return self.values.pop(0)

    # product = v * n; the result is the high word, redrawn while the low word falls under 2^32 mod n
    assert olm.uniform_int_below(Fixed([0xFFFFFFFF]), 10) == 9
    assert olm.uniform_int_below(Fixed([0]), 1) == 0
    n = 3  # threshold = 2^32 mod 3 = 1: low word 0 is rejected
    assert olm.uniform_int_below(Fixed([0, 0x80000000]), n) == 1
    counts = np.zeros(7, dtype=int)
    g = olm.MT19937(99)
    for _ in range(7000):
        counts[olm.uniform_int_below(g, 7)] += 1
    assert counts.min() > 850 and counts.max() < 1150
    idxs = np.arange(100, 130)
    picks = {olm.reference_index(idxs, s) for s in range(200)}
    assert picks <= set(idxs.tolist()) and len(picks) > 20
    assert olm.reference_index(idxs, 2022) == olm.reference_index(idxs, 2022)
    assert olm.reference_index(np.array([17]), 5) == 17


def test_selection_probabilities_follow_the_flat_bottom_energy():
    rng = np.random.default_rng(3)
    box = np.eye(3) * 4.0
    x = rng.random((500, 3)) * 4.0
    radius, k, temperature = 0.9, 3000.0, 300.0
    p = olm.selection_probabilities(x, box, 7, radius, k, temperature)
    d = x - x[7]
    d -= 4.0 * np.rint(d / 4.0)
    r = np.linalg.norm(d, axis=1)
    want = np.exp(-olm.flat_bottom_energy(k, r, 0.0, radius) / (olm.BOLTZ * temperature))
    np.testing.assert_allclose(p, want, rtol=2e-4, atol=1e-7)  # f32 geometry
    assert np.all(p[r < radius] == 1.0) and p[7] == 1.0
    assert np.all(np.diff(p[np.argsort(r)]) <= 1e-6)  # non-increasing with the distance
    # the uniforms: (0, 1], distinct per atom and per seed, mean 1/2
    u = np.array([olm.selection_uniform(5, i) for i in range(4000)])
    assert u.min() > 0 and u.max() <= 1 and abs(u.mean() - 0.5) < 0.02
    assert olm.selection_uniform(5, 1) != olm.selection_uniform(6, 1)
    free, margin = olm.select_free(x, box, 7, radius, k, temperature, seed=5)
    assert not free[7] and np.all(free[(r < radius) & (np.arange(500) != 7)])
    free2, _ = olm.select_free(x, box, 7, radius, k, temperature, seed=5, freeze_reference=False)
    assert free2[7] and np.array_equal(np.delete(free, 7), np.delete(free2, 7))
