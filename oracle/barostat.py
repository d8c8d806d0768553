"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's Monte Carlo barostat arithmetic
(cpp/src/kernels/k_barostat.cuh:10-189, cpp/src/barostat.cu:157-246): the proposal (volume change, molecular centroid
scaling, wrap into the scaled home box) and the Metropolis decision, in the same precision the device code uses.

The uniforms of an attempt are not the reference's (cuRAND, third party, unpinned): the build draws them from
Philox4x32-10 keyed on (seed; attempt), restated here so that a test can follow the GPU attempt by attempt.
"""
import numpy as np

BOLTZ = 0.008314462618  # cpp/src/constants.hpp:5
AVOGADRO = 6.0221367e23  # cpp/src/constants.hpp:6


def philox4x32_10(counter, key):
    """Salmon et al. SC'11, 10 rounds.  counter: 4 uint32, key: 2 uint32 -> 4 uint32."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (int(c) & 0xFFFFFFFF for c in counter)
    k0, k1 = (int(k) & 0xFFFFFFFF for k in key)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def attempt_uniforms(seed, attempt, real=np.float32):
    """(u1, u2) in (0, 1] of barostat attempt number ``attempt`` (0-based), as the device kernel forms them."""
    seed &= 0xFFFFFFFFFFFFFFFF
    r = philox4x32_10((attempt & 0xFFFFFFFF, attempt >> 32, 0x4241524F, 0x53544154), (seed & 0xFFFFFFFF, seed >> 32))
    return real((r[0] + 1.0) / 4294967296.0), real((r[1] + 1.0) / 4294967296.0)


def propose(x, box, group_idxs, volume_scale, u1, real=np.float32):
    """-> x_proposed, box_proposed, (volume, delta, length_scale) -- k_setup_barostat_move + k_find_group_centroids +
    k_rescale_positions.  ``volume_scale`` = 0 means "1 % of the volume" (adaptive first attempt)."""
    from .fixed_point import fixed_to_float, float_to_fixed

    r = real
    volume = r(box[0, 0] * box[1, 1] * box[2, 2])
    if volume_scale == 0.0:
        volume_scale = 0.01 * float(volume)
    delta = r(volume_scale * 2 * float(r(u1) - r(0.5)))
    new_volume = r(volume + delta)
    scale = r(np.cbrt(r(new_volume / volume)))
    box_p = box.copy()
    for d in range(3):
        box_p[d, d] = box[d, d] * float(scale)
    x_p = x.copy()
    for atoms in group_idxs:
        atoms = np.sort(np.asarray(atoms))
        n = r(len(atoms))
        for d in range(3):
            edge = r(box[d, d])
            centre = r(edge * r(0.5))
            total = np.uint64(0)
            with np.errstate(over="ignore"):
                for a in atoms:
                    total = np.uint64(total + float_to_fixed(np.array([x[a, d]]), real=r)[0])
            c = r(r(fixed_to_float(np.array([total]))[0]) / n)
            disp = r(r(r(r(c - centre) * scale) + centre) - c)
            c = r(c + disp)
            sedge = r(edge * scale)
            home = r(sedge * r(np.floor(r(c / sedge))))
            x_p[atoms, d] += float(r(disp - home))
    return x_p, box_p, (float(volume), float(delta), float(scale))


def accept(u_init, u_final, volume, delta, num_molecules, temperature, pressure_bar, u2, real=np.float32):
    """k_decide_move's Metropolis test.  Energies in kJ/mol (None = fixed-point overflow)."""
    r = real
    kT = BOLTZ * float(r(temperature))
    pressure = float(r(pressure_bar)) * AVOGADRO * 1e-25
    energy_delta = np.inf if (u_init is None or u_final is None) else float(r(u_final - u_init))
    new_volume = r(r(volume) + r(delta))
    w = float(r(energy_delta + pressure * float(r(delta)) - num_molecules * kT * np.log(float(new_volume) / float(r(volume)))))
    rejected = w > 0 and float(r(u2)) > float(r(np.exp(-w / kT)))
    return not rejected, w
