"""TEST INFRASTRUCTURE ONLY.  CPU restatement of how local MD picks its atoms
(cpp/src/local_md_potentials.cu:106-147 setup_from_idxs, :179-321 _setup_free_idxs_given_reference_idx;
cpp/src/kernels/k_flat_bottom_bond.cuh:6-80 compute_flat_bottom_energy / k_log_probability_selection).

Pinning:
 * the reference atom is drawn by ``std::mt19937`` + ``std::uniform_int_distribution<unsigned>`` -- both restated here
   (MT19937 as published by Matsumoto & Nishimura; the distribution as libstdc++ 11 implements it, Lemire's nearly
   divisionless reduction, bits/uniform_int_dist.h:243-268) and checked in tests against MT19937's published 10000th output
   and against ``numpy.random.RandomState`` (same generator, same seeding);
 * the per-atom uniforms are cuRAND XORWOW in the reference (third party, absent: *parity unpinned*); the build draws them
   from Philox4x32-10 keyed on the call's seed, restated here so a test can predict the device's selection atom by atom;
 * the acceptance probability follows the reference kernel's arithmetic (f32 geometry, f64 Boltzmann factor).
"""
import numpy as np

from .barostat import philox4x32_10

BOLTZ = 0.008314462618  # cpp/src/constants.hpp:5


class MT19937:
    """std::mt19937 (32-bit Mersenne twister), seeded like ``std::mt19937::seed(value)``."""

    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        self.idx = 0

    def __call__(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def uniform_int_below(rng, n):
    """libstdc++ 11 ``uniform_int_distribution<unsigned>(0, n - 1)(rng)`` for a 32-bit generator (Lemire)."""
    product = rng() * n
    low = product & 0xFFFFFFFF
    if low < n:
        threshold = ((1 << 32) - n) % n
        while low < threshold:
            product = rng() * n
            low = product & 0xFFFFFFFF
    return product >> 32


def reference_index(local_idxs, seed):
    """the atom local_md_potentials.cu:128-132 freezes: local_idxs[uniform_int(0, len - 1)(mt19937(seed))]"""
    return int(local_idxs[uniform_int_below(MT19937(seed), len(local_idxs))])


def selection_uniform(seed, atom):
    """(0, 1] uniform of atom ``atom`` in a selection seeded ``seed`` (timemachine_amd/csrc/local_md.hip: local_md_uniform)."""
    r = philox4x32_10((atom, 0, 0x4C4F4341, 0x4C4D4421), (seed & 0xFFFFFFFF, 0x53454C45))
    return np.float32(((r[0] >> 8) + 1) * 2.0**-24)


def flat_bottom_energy(k, r, rmin, rmax):
    """compute_flat_bottom_energy (k_flat_bottom_bond.cuh:6-20): k/4 [(r - rmin)^4 below rmin + (r - rmax)^4 above rmax]"""
    r = np.asarray(r, dtype=np.float64)
    return (k / 4.0) * (np.where(r < rmin, (r - rmin) ** 4, 0.0) + np.where(r > rmax, (r - rmax) ** 4, 0.0))


def selection_probabilities(x, box, reference_idx, radius, k, temperature):
    """float32 [N]: exp(-U_flat_bottom(|x_i - x_ref|_pbc; 0, radius) / kT), 1 inside the radius."""
    f = np.float32
    x = np.asarray(x, dtype=np.float64)
    radius, k = f(radius), f(k)
    d2 = np.zeros(len(x), dtype=f)
    for d in range(3):
        b = f(box[d, d])
        inv_b = f(1) / b
        delta = (x[:, d] - x[reference_idx, d]).astype(f)
        delta = delta - b * np.rint(delta * inv_b).astype(f)
        d2 = d2 + delta * delta
    dr = np.sqrt(d2) - radius
    dr2 = dr * dr
    energy = (k / f(4.0)) * (dr2 * dr2)
    prob = np.exp(-energy.astype(np.float64) / (BOLTZ * temperature)).astype(f)
    return np.where(d2 >= radius * radius, prob, f(1.0)).astype(f)


def select_free(x, box, reference_idx, radius, k, temperature, seed, freeze_reference=True):
    """-> (free bool[N], margin float[N]): atom i moves iff p_i >= u_i (the reference atom only when it is not frozen);
    margin = |p_i - u_i| lets a test skip atoms whose decision hangs on the last bit of an exponential."""
    p = selection_probabilities(x, box, reference_idx, radius, k, temperature)
    u = np.array([selection_uniform(seed, i) for i in range(len(x))], dtype=np.float32)
    free = p >= u
    free[reference_idx] = not freeze_reference
    return free, np.abs(p.astype(np.float64) - u.astype(np.float64))
