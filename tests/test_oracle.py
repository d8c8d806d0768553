"""CPU tests: the oracle against the committed golden vectors (generated from the reference itself by
tests/golden/generate_golden.py) and its integer models against their definitions.  No GPU, no /root/reference."""
import hashlib
import os

import numpy as np
import pytest

from oracle import fixed_point as fp
from oracle import hilbert as ohilbert
from oracle import integrator as ointegrator
from oracle import nblist as onblist
from oracle import ref_potentials as rp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.mark.parametrize("name", ["nb_small_w0", "nb_small_wrand", "nb_small_whalf"])
def test_nonbonded_energy_matches_reference_golden(name):
    g = load(name + ".npz")
    u, gx, gp = rp.nonbonded(g["x"], g["params"], g["box"], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"]))
    np.testing.assert_allclose(u, float(g["u"]), rtol=1e-12)  # g["u"] was computed by the reference's own nonbonded()
    np.testing.assert_allclose(gx, g["du_dx"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gp, g["du_dp"], rtol=1e-10, atol=1e-10)
    u_ap, _, _ = rp.nonbonded_all_pairs(g["x"], g["params"], g["box"], float(g["beta"]), float(g["cutoff"]))
    np.testing.assert_allclose(u_ap, float(g["u_all_pairs"]), rtol=1e-12)
    u_pl, _, _ = rp.nonbonded_pair_list(g["x"], g["params"], g["box"], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"]))
    np.testing.assert_allclose(u_pl, float(g["u_pair_list"]), rtol=1e-12)
    # decomposition identity the GPU relies on: Nonbonded == AllPairs - sum(scale * pair)
    np.testing.assert_allclose(u, u_ap - u_pl, rtol=1e-12)
    u_sub, _, _ = rp.nonbonded(g["x"], g["params"], g["box"], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"]), atom_idxs=g["atom_idxs"])
    np.testing.assert_allclose(u_sub, float(g["u_subset"]), rtol=1e-12)


def test_bonded_energy_matches_reference_golden():
    g = load("bonded.npz")
    for key, fn, idx, prm in (
        ("bond", rp.harmonic_bond, g["bond_idxs"], g["bond_params"]),
        ("angle", rp.harmonic_angle, g["angle_idxs"], g["angle_params"]),
        ("torsion", rp.periodic_torsion, g["torsion_idxs"], g["torsion_params"]),
    ):
        u, gx, gp = fn(g["x"], prm, g["box"], idx)
        np.testing.assert_allclose(u, float(g[f"u_{key}"]), rtol=1e-12)
        np.testing.assert_allclose(gx, g[f"du_dx_{key}"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gp, g[f"du_dp_{key}"], rtol=1e-10, atol=1e-10)


def test_integrator_matches_reference_golden():
    g = load("integrator.npz")
    x0, v0, masses, k = g["x0"], g["v0"], g["masses"], float(g["k"])
    for friction in (0.0, 1.0):
        ca, cb, cc = ointegrator.langevin_coefficients(300.0, 2.5e-3, friction, masses)
        x, v = x0.copy(), v0.copy()
        for step in range(12):
            force = -k * (x - x0) - 3.0 * (x - x0) ** 3
            x, v = ointegrator.baoab_step(x, v, force, g[f"noise_f{friction:.0f}"][step], ca, cb, cc, 2.5e-3)
            np.testing.assert_array_equal(x, g[f"xs_f{friction:.0f}"][step])  # reference LangevinIntegrator._step, bitwise
            np.testing.assert_array_equal(v, g[f"vs_f{friction:.0f}"][step])


def test_hilbert_lut_and_permutation_match_reference_golden():
    g = load("hilbert.npz")
    lut = ohilbert.lut()
    assert hashlib.sha256(lut.tobytes()).hexdigest() == str(g["lut_sha256"])  # LUT of the reference's vendored C code
    np.testing.assert_array_equal(lut[g["lut_sample_idx"]], g["lut_sample_val"])
    assert int(ohilbert.c2i_3d(np.array([1]), np.array([2]), np.array([3]))[0]) == 36  # SURVEY appendix A spot values
    assert int(ohilbert.c2i_3d(np.array([127]), np.array([127]), np.array([127]))[0]) == 1414745
    water = np.load(os.path.join(GOLDEN, "water.npy"))[:, :3]
    np.testing.assert_array_equal(ohilbert.keys(water, g["box"]), g["keys"])
    np.testing.assert_array_equal(ohilbert.sort_perm(water, g["box"]), g["perm"])


def test_hilbert_ref_build_if_present():
    """oracle/_ref is built from the reference's vendored C where it lies; when present, pin the whole LUT to it."""
    import ctypes

    path = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libhilbert_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (no reference tree)")
    lib = ctypes.CDLL(path)
    out = np.zeros(128**3, dtype=np.uint32)
    lib.ref_hilbert_lut(128, 8, out.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_array_equal(out, ohilbert.lut())


def test_fixed_point_model():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.normal(size=1000) * 1e3, [0.0, -0.0, 0.5 / 2**36, 1.5 / 2**36, -0.5 / 2**36, 2.5 / 2**36]])
    f = fp.float_to_fixed(v)
    # round-half-even at exact halves
    assert int(fp.float_to_fixed(np.array([0.5 / 2**36]))[0]) == 0
    assert int(fp.float_to_fixed(np.array([1.5 / 2**36]))[0]) == 2
    assert int(fp.float_to_fixed(np.array([2.5 / 2**36]))[0]) == 2
    # negation symmetry: FIX(-v) == -FIX(v) in two's complement (what makes g_i + g_j == 0 exactly)
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(fp.float_to_fixed(-v), (np.uint64(0) - f))
    np.testing.assert_allclose(fp.fixed_to_float(f), v, atol=2.0**-37)
    # f32 products are formed in f32
    x32 = np.float32(1.0) / np.float32(3.0)
    assert int(fp.float_to_fixed(x32, real=np.float32)[()]) == int(np.rint(np.float64(np.float32(x32 * np.float32(2**36)))))
    # energies: overflow clamps to LLONG_MAX and the sum of two clamps is detected
    assert fp.float_to_fixed_energy(1e30) == fp.LLONG_MAX
    assert fp.float_to_fixed_energy(float("inf")) == fp.LLONG_MAX
    assert fp.float_to_fixed_energy(-1e30) == fp.LLONG_MAX
    assert fp.fixed_point_overflow(int(fp.LLONG_MAX) - 1 + int(fp.LLONG_MAX) - 1)
    assert np.isnan(fp.energy_to_float(int(fp.LLONG_MAX)))
    assert fp.energy_to_float(fp.float_to_fixed_energy(1.25)) == 1.25
    # per-column du_dp exponents
    d = np.array([[1 << 36, 1 << 37, 1 << 38, 1 << 36]], dtype=np.uint64)
    np.testing.assert_array_equal(fp.nb_du_dp_fixed_to_float(d), np.ones((1, 4)))


def _reference_block_bounds_numpy(coords, box, block_size):
    """the reference test-suite's own numpy model (tests/test_nblist.py:28-56), restated"""
    coords = coords.copy()
    N = coords.shape[0]
    nb = (N + block_size - 1) // block_size
    bd = np.diagonal(box)
    ctrs, exts = [], []
    for b in range(nb):
        blk = coords[b * block_size : min((b + 1) * block_size, N)]
        lo = blk[0]
        hi = blk[0]
        for c in blk[1:]:
            ctr = 0.5 * (hi + lo)
            c = c - bd * np.floor((c - ctr) / bd + 0.5)
            lo = np.minimum(lo, c)
            hi = np.maximum(hi, c)
        ctrs.append((hi + lo) / 2)
        exts.append((hi - lo) / 2)
    return np.array(ctrs), np.array(exts)


@pytest.mark.parametrize("size", [12, 128, 156, 298])
def test_block_bounds_model(size):
    rng = np.random.default_rng(2020)
    coords = rng.normal(size=(size, 3))
    box = np.eye(3) * (rng.uniform(size=3) + 1)
    ref_c, ref_e = _reference_block_bounds_numpy(coords, box, 32)
    c, e = onblist.block_bounds(coords, box)
    np.testing.assert_allclose(c, ref_c, atol=1e-7, rtol=1e-7)  # tolerance of tests/test_nblist.py:58
    np.testing.assert_allclose(e, ref_e, atol=1e-7, rtol=1e-7)


def test_brute_force_ixn_list_is_symmetric_cover():
    water = np.load(os.path.join(GOLDEN, "water.npy"))[:300, :3]
    box = np.eye(3) * (water.max(0) - water.min(0) + 0.1)
    lst = onblist.brute_force_ixn_list(water, box, 1.0)
    # every pair within the cutoff appears in the row block of its smaller index
    b = np.diagonal(box)
    d = water[:, None, :] - water[None, :, :]
    d -= b * np.floor(d / b + 0.5)
    within = np.linalg.norm(d, axis=-1) < 1.0
    for i in range(300):
        for j in np.nonzero(within[i])[0]:
            if j >= (i // 32) * 32:
                assert j in lst[i // 32]


def test_groups_precomputed_chiral_match_reference_golden():
    """the expected energies in groups.npz came from the reference's nonbonded_interaction_groups,
    nonbonded_on_precomputed_pairs, chiral_atom_restraint and chiral_bond_restraint."""
    g = load("groups.npz")
    beta, cutoff = float(g["beta"]), float(g["cutoff"])
    for tag, cols in (("ig_all", None), ("ig_sub", g["ig_cols_sub"])):
        u, gx, gp = rp.nonbonded_interaction_group(g["ig_x"], g["ig_params"], g["ig_box"], g["ig_rows"], beta, cutoff, cols)
        np.testing.assert_allclose(u, float(g[f"{tag}_u"]), rtol=1e-12)
        np.testing.assert_allclose(gx, g[f"{tag}_du_dx"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gp, g[f"{tag}_du_dp"], rtol=1e-10, atol=1e-10)
    # an interaction group is the all-pairs energy of the union minus the two within-group energies
    rows, cols = g["ig_rows"], g["ig_cols_sub"]
    union = np.sort(np.concatenate([rows, cols])).astype(np.int32)
    u_union, _, _ = rp.nonbonded_all_pairs(g["ig_x"], g["ig_params"], g["ig_box"], beta, cutoff, atom_idxs=union)
    u_rows, _, _ = rp.nonbonded_all_pairs(g["ig_x"], g["ig_params"], g["ig_box"], beta, cutoff, atom_idxs=rows)
    u_cols, _, _ = rp.nonbonded_all_pairs(g["ig_x"], g["ig_params"], g["ig_box"], beta, cutoff, atom_idxs=cols)
    np.testing.assert_allclose(u_union - u_rows - u_cols, float(g["ig_sub_u"]), rtol=1e-10)
    u, gx, gp = rp.nonbonded_pair_list_precomputed(g["ig_x"], g["pre_params"], g["ig_box"], g["pre_idxs"], beta, cutoff)
    np.testing.assert_allclose(u, float(g["pre_u"]), rtol=1e-12)
    np.testing.assert_allclose(gx, g["pre_du_dx"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gp, g["pre_du_dp"], rtol=1e-10, atol=1e-10)
    u, gx, gp = rp.chiral_atom_restraint(g["chiral_x"], g["chiral_atom_params"], None, g["chiral_atom_idxs"])
    np.testing.assert_allclose(u, float(g["chiral_atom_u"]), rtol=1e-12)
    np.testing.assert_allclose(gx, g["chiral_atom_du_dx"], rtol=1e-10, atol=1e-10)
    u, gx, gp = rp.chiral_bond_restraint(g["chiral_x"], g["chiral_bond_params"], None, g["chiral_bond_idxs"], g["chiral_bond_signs"])
    np.testing.assert_allclose(u, float(g["chiral_bond_u"]), rtol=1e-12)
    np.testing.assert_allclose(gx, g["chiral_bond_du_dx"], rtol=1e-10, atol=1e-10)


def test_restraint_potentials_match_reference_golden():
    """flat-bottom / log flat-bottom bonds and the centroid restraint (bonded.py:8-31,219-253)"""
    g = load("groups.npz")
    x = g["chiral_x"]
    u, gx, gp = rp.flat_bottom_bond(x, g["fb_params"], g["fb_box"], g["fb_idxs"])
    np.testing.assert_allclose(u, float(g["fb_u"]), rtol=1e-12)
    np.testing.assert_allclose(gx, g["fb_du_dx"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gp, g["fb_du_dp"], rtol=1e-10, atol=1e-10)
    u, gx, gp = rp.log_flat_bottom_bond(x, g["lfb_params"], g["fb_box"], g["lfb_idxs"], float(g["lfb_beta"]))
    np.testing.assert_allclose(u, float(g["lfb_u"]), rtol=1e-12)
    np.testing.assert_allclose(gx, g["lfb_du_dx"], rtol=1e-10, atol=1e-10)
    for tag in ("cr", "cr0"):
        u, gx, _ = rp.centroid_restraint(x, None, None, g["cr_a"], g["cr_b"], float(g[f"{tag}_kb"]), float(g[f"{tag}_b0"]))
        np.testing.assert_allclose(u, float(g[f"{tag}_u"]), rtol=1e-12)
        np.testing.assert_allclose(gx, g[f"{tag}_du_dx"], rtol=1e-10, atol=1e-10)


# ----------------------------------------------------------------------------------------------------------------
# The "next" rows: oracles pinned to fixtures made by the reference's own Python
# (tests/golden/generate_golden_next.py imports timemachine/integrator.py, md/barostat/moves.py, md/hrex.py)
# ----------------------------------------------------------------------------------------------------------------
def test_velocity_verlet_oracle_matches_reference_fixture():
    """oracle/integrator.py:velocity_verlet_device_model (double state, k_integrator.cuh:64-130) against the trajectory of
    the reference's Python VelocityVerletIntegrator.multiple_steps (fixed-point state, timemachine/integrator.py:169-199).
    The two differ by the 2^-36 quantisation of the reference's state only."""
    from oracle import integrator as oi
    from oracle import ref_potentials as rp
    from oracle.fixed_point import float_to_fixed

    g = load("vv.npz")
    box = g["box"]

    def grad_fixed(x):
        gb = rp.harmonic_bond(x, g["bond_params"], box, g["bond_idxs"])[1]
        ga = rp.harmonic_angle(x, g["angle_params"], box, g["angle_idxs"])[1]
        return float_to_fixed(gb + ga)

    dt, n_steps = float(g["dt"]), int(g["n_steps"])
    x, v = oi.velocity_verlet_device_model(g["x0"], g["v0"], grad_fixed, -dt / g["masses"], dt, n_steps + 1)
    # (the generator's own bound: the 2^-36 quantisation of the reference's state through 11 steps of stiff O-H dynamics)
    np.testing.assert_allclose(x, g["ref_xs"][-1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v, g["ref_vs"][-1], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g["ref_xs"][0], g["x0"], rtol=0, atol=2.0**-36)  # zs[0] is the start, quantised (integrator.py:179-181)
    # frame by frame: the reference stores x after k + 1 drifts in xs[k] (k >= 1; "xs[1] = x_2", integrator.py:171-177)
    for k in (2, 5, n_steps + 1):
        xk, _ = oi.velocity_verlet_device_model(g["x0"], g["v0"], grad_fixed, -dt / g["masses"], dt, k)
        np.testing.assert_allclose(xk, g["ref_xs"][k - 1], rtol=0, atol=1e-9)


def test_barostat_oracle_matches_reference_centroid_rescaler():
    """oracle/barostat.py:propose against the reference's CentroidRescaler.scale_centroids (md/barostat/moves.py:39-83)
    evaluated at the oracle's own f32 length scales: equal modulo the wrap into the scaled home box that the device adds
    (whole molecules shifted by whole box edges), to f32 accuracy; the f64 form is asserted to 1e-9 by the generator."""
    from oracle import barostat as ob

    g = load("barostat.npz")
    x, box = g["x"], g["box"]
    bounds = np.concatenate([[0], np.cumsum(g["group_sizes"])])
    groups = [np.arange(bounds[k], bounds[k + 1]) for k in range(len(bounds) - 1)]
    for attempt in range(len(g["scales"])):
        u1, _ = ob.attempt_uniforms(int(g["seed"]), attempt)
        x_p, box_p, (_, _, scale) = ob.propose(x, box, groups, float(g["volume_scale"]), u1, real=np.float32)
        assert scale == g["scales"][attempt]
        shift = (x_p - g["x_scaled"][attempt]) / np.diagonal(box_p)
        resid = (shift - np.rint(shift)) * np.diagonal(box_p)
        assert np.abs(resid).max() < 5e-6, (attempt, np.abs(resid).max())
        for grp in groups:
            assert np.all(np.rint(shift[grp]) == np.rint(shift[grp][0]))


def test_hrex_swap_chain_matches_reference_fixture():
    """timemachine_amd.hrex.run_neighbor_swaps and oracle/hrex.py against the reference's _run_neighbor_swaps
    (timemachine/md/hrex.py:50-130) on recorded pair indices / uniforms, incl. -inf (unevaluated) entries: bitwise."""
    from oracle import hrex as oh
    from timemachine_amd import hrex as th

    g = load("hrex.npz")
    for tag in ("a", "b"):
        args = (g[f"{tag}_perm0"], g[f"{tag}_pairs"], g[f"{tag}_log_q"], g[f"{tag}_pair_idxs"], g[f"{tag}_uniforms"])
        perm, proposed, accepted = th.run_neighbor_swaps(*args)
        for got, want in zip(th.run_neighbor_swaps_python(*args), (perm, proposed, accepted)):
            np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(perm, g[f"{tag}_perm"])
        np.testing.assert_array_equal(proposed, g[f"{tag}_proposed"])
        np.testing.assert_array_equal(accepted, g[f"{tag}_accepted"])
        p2, pr2, ac2 = oh.run_moves(list(args[0]), [tuple(p) for p in args[1]], args[2], args[3], args[4])
        np.testing.assert_array_equal(p2, g[f"{tag}_perm"])
        np.testing.assert_array_equal(ac2, g[f"{tag}_accepted"])
        assert int(g[f"{tag}_accepted"].sum()) > 0


def test_edge_case_goldens_against_oracle():
    """orthorhombic box / whole-box drifts / config 1 in both boxes: the oracle reproduces the recorded reference energies
    (so GPU tests that compare with these fixtures and with the oracle on the fly are comparing with the same thing)."""
    from oracle import ref_potentials as rp

    for name in ("edge_ortho", "edge_drift", "config1_vacuum", "config1_pbc"):
        g = load(name + ".npz")
        u, gx, _ = rp.nonbonded(g["x"], g["params"], g["box"], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"]))
        assert abs(u - float(g["u"])) <= 1e-11 * max(1.0, abs(u))
        np.testing.assert_allclose(gx, g["du_dx"], rtol=0, atol=1e-9)
    a, b = load("edge_ortho.npz"), load("edge_drift.npz")
    assert abs(float(a["u"]) - float(b["u"])) < 1e-9 * abs(float(a["u"]))
