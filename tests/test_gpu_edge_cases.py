"""GPU parity tests, part 2 (``-m gpu``): the reference's edge cases, BASELINE configs 1 / 4 / 5 at their stated
sizes, the "next" rows (VelocityVerlet, barostat, HREX) against fixtures made by the reference's own Python, and the
device fixed-point conversions bit for bit.  Every call goes Python -> ctypes -> C ABI -> HIP kernels.

Fixtures: tests/golden/*.npz written by tests/golden/generate_golden_next.py in the build container (reference energies;
oracle gradients after energy-equality + finite-difference checks against the reference).
"""
import os

import numpy as np
import pytest

from test_gpu_parity import TOL, assert_equal_vectors, compare_forces, load  # noqa: F401  (same bars, same helpers)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    assert custom_ops.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return custom_ops


@pytest.fixture(scope="module")
def P():
    from timemachine_amd import potentials

    return potentials


# ----------------------------------------------------------------------------------------------------------------
# a3: the device's fixed-point conversions, bit for bit (k_fixed_point.cuh:10-98, fixed_point.hpp:5-34)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_device_fixed_point_conversions_bit_exact(co, precision):
    from oracle import fixed_point as fp

    rng = np.random.default_rng(11)
    n = 1_000_000
    mag = np.exp(rng.uniform(np.log(1e-13), np.log(6e7), n))  # |v * 2^36| from far below 1 up to ~2^62
    v = mag * rng.choice([-1.0, 1.0], n)
    half = (rng.integers(-(1 << 20), 1 << 20, 4096) + 0.5) / 2.0**36  # exact ties: round half to even
    edge = np.array([0.0, -0.0, 1.0, -1.0, 2.0**-37, -(2.0**-37), 3 * 2.0**-37, (2.0**51 - 1) / 2.0**36, -(2.0**51 - 1) / 2.0**36,
                     2.0**15, -(2.0**15), (2.0**51 + 1) / 2.0**36, 2.0**25, -(2.0**25), 2.0**26 * (1 - 2.0**-40), -(2.0**26) * (1 - 2.0**-40)])
    vals = np.concatenate([v, half, edge])
    if precision == np.float32:
        vals = vals.astype(np.float32).astype(np.float64)
    for kind, exponent in ((0, fp.FIXED_EXPONENT), (1, fp.FIXED_EXPONENT_DU_DSIG), (2, fp.FIXED_EXPONENT_DU_DEPS)):
        x = vals[np.abs(vals) * exponent < 2.0**62]  # beyond the int64 range llrint is undefined in the reference too
        got = co.debug_float_to_fixed(x, precision, kind)
        np.testing.assert_array_equal(got, fp.float_to_fixed(x, exponent, real=precision))
    # the nonbonded force form FIX(prefactor * delta): f64 product in f64, f32 product ROUNDED to f32 (k_nonbonded.cuh:248-254)
    pre = np.exp(rng.uniform(np.log(1e-6), np.log(1e7), n)) * rng.choice([-1.0, 1.0], n)
    dlt = rng.uniform(-1.3, 1.3, n)
    pairs = np.stack([pre, dlt], 1).astype(precision).astype(np.float64)
    prod = (pairs[:, 0].astype(precision) * pairs[:, 1].astype(precision)).astype(np.float64)
    keep = np.abs(prod) < 2.0**25
    got = co.debug_float_to_fixed(pairs[keep], precision, 3)
    np.testing.assert_array_equal(got, fp.float_to_fixed(prod[keep], fp.FIXED_EXPONENT, real=np.float64))
    # energies: clamp non-finite and out-of-range values to LLONG_MAX (k_fixed_point.cuh:88-98)
    ev = np.concatenate([rng.normal(size=2000) * 1e3, [0.0, np.nan, np.inf, -np.inf, 2.0**27, -(2.0**27), 2.0**27 * (1 - 2.0**-30), -(2.0**27) * (1 - 2.0**-30), 1e300, -1e300, 134217727.5]])
    if precision == np.float32:
        ev = ev.astype(np.float32).astype(np.float64)
    got = co.debug_float_to_fixed_energy(ev, precision)
    want = [fp.float_to_fixed_energy(e, real=precision) for e in ev]
    assert got == want
    assert got[2001] == got[2002] == got[2003] == fp.LLONG_MAX


# ----------------------------------------------------------------------------------------------------------------
# f4: VelocityVerletIntegrator against the reference's Python integrator (fixture vv.npz)
# ----------------------------------------------------------------------------------------------------------------
def test_velocity_verlet_matches_reference_python_trajectory(co, P):
    """tests/test_velocity_verlet_integrator.py:104-140 (test_matches_reference): Context.multiple_steps(n, 1) frames ==
    ReferenceVelocityVerlet.multiple_steps(n + 1)[1:-1], final velocities == its last; the reference's bar is atol 1e-5
    (f32 potentials); with f64 potentials only the reference's 2^-36 state quantisation separates the two."""
    from timemachine_amd.lib import VelocityVerletIntegrator

    g = load("vv.npz")
    n_steps = int(g["n_steps"])
    bps = [
        P.HarmonicBond(g["bond_idxs"]).bind(g["bond_params"]).to_gpu(np.float64).bound_impl,
        P.HarmonicAngle(g["angle_idxs"]).bind(g["angle_params"]).to_gpu(np.float64).bound_impl,
    ]
    ctxt = co.Context(g["x0"], g["v0"], g["box"], VelocityVerletIntegrator(float(g["dt"]), g["masses"]).impl(), bps)
    xs, boxes = ctxt.multiple_steps(n_steps, 1)
    assert xs.shape[0] == n_steps
    np.testing.assert_allclose(xs, g["ref_xs"][1:-1], rtol=0, atol=2e-9)
    np.testing.assert_allclose(ctxt.get_v_t(), g["ref_vs"][-1], rtol=0, atol=2e-7)
    # f32 potentials, the reference's own tolerance
    bps32 = [
        P.HarmonicBond(g["bond_idxs"]).bind(g["bond_params"]).to_gpu(np.float32).bound_impl,
        P.HarmonicAngle(g["angle_idxs"]).bind(g["angle_params"]).to_gpu(np.float32).bound_impl,
    ]
    c32 = co.Context(g["x0"], g["v0"], g["box"], VelocityVerletIntegrator(float(g["dt"]), g["masses"]).impl(), bps32)
    xs32, _ = c32.multiple_steps(n_steps, 1)
    np.testing.assert_allclose(xs32, g["ref_xs"][1:-1], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c32.get_v_t(), g["ref_vs"][-1], rtol=0, atol=1e-5 * 100)


# ----------------------------------------------------------------------------------------------------------------
# f2: barostat proposal against the reference's CentroidRescaler (fixture barostat.npz)
# ----------------------------------------------------------------------------------------------------------------
def test_barostat_accepted_moves_are_the_reference_centroid_scaling(co, P):
    """Every accepted MonteCarloBarostat move equals timemachine/md/barostat/moves.py:CentroidRescaler.scale_centroids at
    the move's length scale, modulo the wrap of whole molecules into the scaled home box (k_barostat.cuh).  Only the
    intramolecular bonds are bound, so dU = 0 and the Metropolis factor is the ideal-gas one: expansions are always
    accepted (w = P dV - N kT ln(V'/V) < 0)."""
    from timemachine_amd.lib import MonteCarloBarostat

    g = load("barostat.npz")
    x, box = g["x"], g["box"]
    bounds = np.concatenate([[0], np.cumsum(g["group_sizes"])])
    groups = [list(range(bounds[k], bounds[k + 1])) for k in range(len(bounds) - 1)]
    N = x.shape[0]
    bps = [P.HarmonicBond(g["bond_idxs"]).bind(g["bond_params"]).to_gpu(np.float32).bound_impl]
    baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 1, int(g["seed"]), adaptive_scaling_enabled=False, initial_volume_scale_factor=0.0).impl(bps)
    baro.set_volume_scale_factor(float(g["volume_scale"]))
    n_accept = 0
    for attempt, scale in enumerate(g["scales"]):
        x_new, box_new = baro.move(x, box)  # always from the fixture's state: attempt k uses the k-th uniforms
        if np.array_equal(box_new, box):
            np.testing.assert_array_equal(x_new, x)
            assert scale < 1.0, "an expansion of an ideal gas of molecules must be accepted"
            continue
        n_accept += 1
        np.testing.assert_allclose(np.diagonal(box_new), np.diagonal(box) * scale, rtol=1e-6)
        shift = (x_new - g["x_scaled"][attempt]) / np.diagonal(box_new)
        resid = (shift - np.rint(shift)) * np.diagonal(box_new)
        assert np.abs(resid).max() < 2e-5, (attempt, np.abs(resid).max())
        for grp in groups:
            assert np.all(np.rint(shift[grp]) == np.rint(shift[grp][0]))  # molecules are wrapped whole
        cent = np.array([x_new[grp].mean(0) for grp in groups])
        assert np.all(cent >= -1e-5) and np.all(cent <= np.diagonal(box_new) + 1e-5)
    assert n_accept >= int((g["scales"] > 1.0).sum()) >= 2


# ----------------------------------------------------------------------------------------------------------------
# f3: the HREX energy matrix against the oracle (not against another batch of the same kernels)
# ----------------------------------------------------------------------------------------------------------------
def test_hrex_energy_matrix_matches_oracle(co, P):
    from oracle import ref_potentials as rp
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts

    n_states = 5
    lambdas = np.linspace(0.0, 0.4, n_states)
    systems = [ts.small_solvated_ligand(lamb=float(lam)) for lam in lambdas]
    s0 = systems[0]
    params_by_state = np.stack([s.nb_params for s in systems])
    rng = np.random.default_rng(3)
    coords = np.stack([s0.coords + rng.normal(size=s0.coords.shape) * 0.002 for _ in range(n_states)])
    boxes = np.stack([s0.box] * n_states)
    unbound = P.Nonbonded(s0.num_atoms, s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff).to_gpu(np.float64).unbound_impl
    dh = hrex.DistributedHREX(n_states, 300.0, max_delta_states=1)
    rows = hrex.compute_potential_matrix(unbound, coords, boxes, params_by_state, dh.replica_idx_by_state, 1)
    evaluated = np.argwhere(np.isfinite(rows))
    assert len(evaluated) == 3 * n_states - 2
    for r, s in evaluated:
        u = float(rp.nonbonded_energy(rp._t(coords[r]), rp._t(params_by_state[s]), rp._t(boxes[r]), s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff))
        np.testing.assert_allclose(rows[r, s], u, rtol=1e-8, atol=1e-8)
    # the exchange itself: recorded reference chain (md/hrex.py:_run_neighbor_swaps) through the product function
    g = load("hrex.npz")
    perm, proposed, accepted = hrex.run_neighbor_swaps(g["a_perm0"], g["a_pairs"], g["a_log_q"], g["a_pair_idxs"], g["a_uniforms"])
    np.testing.assert_array_equal(perm, g["a_perm"])
    np.testing.assert_array_equal(accepted, g["a_accepted"])


# ----------------------------------------------------------------------------------------------------------------
# Reference edge cases: orthorhombic box, drifted coordinates, box resize on one impl, order independence, reversal
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["edge_ortho", "edge_drift"])
def test_orthorhombic_box_and_whole_box_drifts(co, P, name, precision):
    """Lx != Ly != Lz (only ortholinear boxes exist in the reference, wrap_kernels.cpp:51-78); and the same atoms moved by up
    to +-3 box vectors each -- MD never re-images coordinates, so this is the normal state of a long run."""
    g = load(name + ".npz")
    x, p, box = g["x"], g["params"], g["box"]
    assert len({box[0, 0], box[1, 1], box[2, 2]}) == 3
    impl = P.Nonbonded(x.shape[0], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"])).to_gpu(precision).unbound_impl
    if precision == np.float32 and name == "edge_drift":
        # f32 kernels see coordinates rounded to f32 (k_nonbonded.cuh:134-151): at |x| ~ 14 nm that is 1e-6 nm of noise
        du_dx, du_dp, u = impl.execute(x, p, box)
        np.testing.assert_allclose(u, float(g["u"]), rtol=2e-3, atol=5e-2)
        assert_equal_vectors(g["du_dx"], du_dx, 5e-3)
    else:
        compare_forces(impl, x, p, box, float(g["u"]), g["du_dx"], g["du_dp"], precision)
    nohilb = P.Nonbonded(x.shape[0], g["exclusion_idxs"], g["scale_factors"], float(g["beta"]), float(g["cutoff"]), disable_hilbert_sort=True).to_gpu(precision).unbound_impl
    a, b = impl.execute_raw(x, p, box), nohilb.execute_raw(x, p, box)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2] == b[2]


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_nblist_box_resize(co, P, precision):
    """tests/nonbonded/test_nonbonded.py:165-190: ONE impl evaluated under box + 1000 I, then under the real box -- the
    list must be rebuilt because the box changed, whatever the coordinates did."""
    g = load("edge_box_resize.npz")
    x, p = g["x"], g["params"]
    impl = P.NonbondedAllPairs(x.shape[0], float(g["beta"]), float(g["cutoff"])).to_gpu(precision).unbound_impl
    t = TOL[precision]
    for tag in ("big", "real", "big"):
        du_dx, du_dp, u = impl.execute(x, p, g[f"box_{tag}"])
        np.testing.assert_allclose(u, float(g[f"u_{tag}"]), rtol=t["rtol"], atol=t["atol"])
        assert_equal_vectors(g[f"du_dx_{tag}"], du_dx, t["rtol"])
        np.testing.assert_allclose(du_dp, g[f"du_dp_{tag}"], rtol=t["prtol"] * 10, atol=t["patol"] * 10)
    assert float(g["u_big"]) != float(g["u_real"])


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("num_atoms_subset", [None, 33])
@pytest.mark.parametrize("num_atoms", [33, 65, 231])
def test_nonbonded_all_pairs_order_independent(co, P, num_atoms, num_atoms_subset, precision):
    """tests/nonbonded/test_nonbonded_all_pairs.py:202-238: with and without the Hilbert sort, bitwise, incl. atom subsets
    and 4D offsets (w in {0, random, cutoff})."""
    g = load("config1_pbc.npz")
    rng = np.random.default_rng(num_atoms)
    x, box = g["x"][:num_atoms], g["box"]
    params = g["params"][:num_atoms].copy()
    beta, cutoff = 2.0, 1.1
    atom_idxs = rng.choice(num_atoms, size=(num_atoms_subset,), replace=False).astype(np.int32) if num_atoms_subset else None
    a = P.NonbondedAllPairs(num_atoms, beta, cutoff, atom_idxs).to_gpu(precision).unbound_impl
    b = P.NonbondedAllPairs(num_atoms, beta, cutoff, atom_idxs, disable_hilbert_sort=True).to_gpu(precision).unbound_impl
    for w in (np.zeros(num_atoms), rng.uniform(-cutoff, cutoff, num_atoms), np.full(num_atoms, cutoff) * (np.arange(num_atoms) % 2)):
        params[:, 3] = w
        ra, rb = a.execute(x, params, box), b.execute(x, params, box)
        np.testing.assert_array_equal(ra[0], rb[0])
        np.testing.assert_array_equal(ra[1], rb[1])
        assert ra[2] == rb[2]


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_torsion_index_reversal_is_bitwise(co, P, precision):
    """PeriodicTorsion (i, j, k, l) -> (l, k, j, i) describes the same term: same energy, forces and du_dp bit for bit
    (Appendix B.11: the no-FMA rmul/rsub forms of k_periodic_torsion.cuh exist for exactly this)."""
    g = load("bonded.npz")
    x, box = g["x"], g["box"]
    idxs, prm = g["torsion_idxs"], g["torsion_params"]
    fwd = P.PeriodicTorsion(idxs).to_gpu(precision).unbound_impl.execute_raw(x, prm, box)
    rev = P.PeriodicTorsion(np.ascontiguousarray(idxs[:, ::-1])).to_gpu(precision).unbound_impl.execute_raw(x, prm, box)
    np.testing.assert_array_equal(fwd[0], rev[0])
    np.testing.assert_array_equal(fwd[1], rev[1])
    assert fwd[2] == rev[2]


@pytest.mark.parametrize("cutoff", [1.0, 1.2, 1.4])
def test_f64_forces_only_forms_agree_with_the_full_call_and_the_oracle(co, P, cutoff):
    """The MD launch of the f64 tile kernel (forces only) has its own pair path: table-driven electrostatics with the rare
    cases deferred, and -- for cutoffs up to the end of the switch at 1.2 nm -- a form without the beyond-the-switch select.
    Below, at and beyond 1.2 nm its forces must equal the full call's (u + du_dx + du_dp, analytic damping function on the
    side) bit for bit, and the oracle's within the f64 bar; beyond 1.2 nm pairs exist whose electrostatic term is exactly 0."""
    from oracle import ref_potentials as rp

    g = load("config1_pbc.npz")
    x, p, box = g["x"], g["params"], g["box"]
    pot = P.NonbondedAllPairs(256, 2.0, cutoff).to_gpu(np.float64).unbound_impl
    only = pot.execute_raw(x, p, box, True, False, False)[0]
    full = pot.execute_raw(x, p, box, True, True, True)[0]
    with_u = pot.execute_raw(x, p, box, True, False, True)[0]
    np.testing.assert_array_equal(only, full)
    np.testing.assert_array_equal(only, with_u)
    u_ref, du_dx_ref, _ = rp.nonbonded_all_pairs(x, p, box, 2.0, cutoff)
    du_dx = pot.execute(x, p, box, True, False, False)[0]
    assert_equal_vectors(du_dx_ref, du_dx, TOL[np.float64]["rtol"])


# ----------------------------------------------------------------------------------------------------------------
# tests/test_energy_overflows.py:68-128,131-177,251-280 -- two atoms in the 100 nm vacuum box; summation overflow
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_two_atom_vacuum_overflows(co, P, precision):
    from oracle import ref_potentials as rp

    box = np.eye(3) * 100.0
    beta, cutoff = 2.0, 1.0
    params = np.array([[-np.sqrt(138.935456), 0.0, 0.0, 0.0], [np.sqrt(138.935456), 0.0, 0.0, 0.0]])  # -1 e / +1 e, no LJ
    nb = P.Nonbonded(2, np.zeros((0, 2), np.int32), np.zeros((0, 2)), beta, cutoff).to_gpu(precision)
    bound = nb.bind(params).bound_impl
    limit = float(np.iinfo(np.int64).max) / co.FIXED_EXPONENT  # 2^27 kJ/mol
    t = TOL[precision]
    saw_overflow = False
    for d in (0.5, 0.1, 1e-2, 1e-4, 1e-6, 1e-7, 1e-8, 1e-9, 1e-12):
        x = np.array([[50.0, 50.0, 50.0], [50.0 + d, 50.0, 50.0]])
        ref_u = float(rp.nonbonded_energy(rp._t(x), rp._t(params), rp._t(box), np.zeros((0, 2), np.int32), np.zeros((0, 2)), beta, cutoff))
        _, _, u = nb.unbound_impl.execute(x, params, box, False, False, True)
        fixed = bound.execute_fixed(x, box)
        if abs(ref_u) < 0.5 * limit and (precision == np.float64 or d > 1e-6):  # f32 coordinates cannot resolve 50 + 1e-7
            # f32 kernels see coordinates rounded to f32 (k_nonbonded.cuh:134-151): at x = 50 nm one ulp is 3.8e-6 nm of d
            np.testing.assert_allclose(u, ref_u, rtol=max(t["rtol"], 8e-6 / d if precision == np.float32 else 0), atol=t["atol"])
            assert int(fixed[0]) != (1 << 63) - 1
        elif abs(ref_u) > 2 * limit and precision == np.float64:
            assert np.isnan(u), (d, u, ref_u)  # the reference stays finite, the fixed-point platform does not
            assert int(fixed[0]) == (1 << 63) - 1
            saw_overflow = True
    assert saw_overflow or precision == np.float32
    # exactly overlapping: the reference potential gives -inf, the GPU platform NaN and an overflowed fixed energy
    x = np.array([[50.0, 50.0, 50.0], [50.0, 50.0, 50.0]])
    _, _, u = nb.unbound_impl.execute(x, params, box, False, False, True)
    assert np.isnan(u)
    assert int(bound.execute_fixed(x, box)[0]) == (1 << 63) - 1


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_energy_overflows_with_summation_of_energies(co, P, precision):
    """tests/test_energy_overflows.py:251-280: 1 000 atoms in a line, every pair energy representable, their sum not."""
    from oracle import ref_potentials as rp

    num_atoms, spacing = 1000, 0.75
    x = spacing * np.array([np.arange(num_atoms), np.zeros(num_atoms), np.zeros(num_atoms)]).T
    box = np.eye(3) * 10000.0
    params = np.ones((num_atoms, 4))
    params[:, -1] = 0.0
    excl, scales = np.array([(0, num_atoms - 1)], dtype=np.int32), np.zeros((1, 2))
    ref_u = float(rp.nonbonded_energy(rp._t(x), rp._t(params), rp._t(box), excl, scales, 2.0, 1.2))
    assert ref_u > np.iinfo(np.int64).max / co.FIXED_EXPONENT
    nb = P.Nonbonded(num_atoms, excl, scales, 2.0, 1.2).to_gpu(precision)
    du_dx, _, u = nb.unbound_impl.execute(x, params, box, True, False, True)
    assert np.isnan(u)
    assert np.all(np.isfinite(du_dx))  # forces wrap, only the energy carries the overflow flag
    assert int(nb.bind(params).bound_impl.execute_fixed(x, box)[0]) == (1 << 63) - 1
    # the same through a SummedPotential of two halves: the children's energies are summed in 128 bits as well
    half = P.SummedPotential([P.Nonbonded(num_atoms, excl, scales, 2.0, 1.2)] * 2, [params, params]).to_gpu(precision)
    _, _, u2 = half.unbound_impl.execute(x, np.concatenate([params.reshape(-1)] * 2), box, False, False, True)
    assert np.isnan(u2)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 1: 85 waters + 1 LJ atom = 256 atoms, 100 nm vacuum box and 3.0 nm periodic box
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("tag", ["vacuum", "pbc"])
def test_config1_both_boxes(co, P, tag, precision, nb_path):
    """In the 100 nm box the Hilbert grid's bins are 0.79 nm wide (the whole cluster sits in a handful of bins) and every
    block bound is a sliver of the box: list build and tile kernel must not care."""
    from timemachine_amd import testsystems as ts

    g = load(f"config1_{tag}.npz")
    s = ts.config1_water_cluster(100.0 if tag == "vacuum" else 3.0)
    np.testing.assert_array_equal(s.box, g["box"])
    x, p, box = g["x"], g["params"], g["box"]
    assert x.shape == (256, 3)
    nb = P.Nonbonded(256, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(precision).unbound_impl
    compare_forces(nb, x, p, box, float(g["u"]), g["du_dx"], g["du_dp"], precision)
    # f32: 1e-4 of the force norm.  The fixture's coordinates are strained (tests/golden/generate_golden.py: strained), so the
    # stiff O-H bonds pull with ~1e3 kJ/mol/nm and their absolute f32 error (k * eps_f32 * r ~ 3e-3 kJ/mol/nm) is 1e-6 of it
    brt = 1e-7 if precision == np.float64 else 1e-4
    for pot, prm, key in ((P.HarmonicBond(s.bond_idxs), s.bond_params, "bond"), (P.HarmonicAngle(s.angle_idxs), s.angle_params, "angle")):
        assert np.linalg.norm(g[f"du_dx_{key}"], axis=1).max() > 100.0  # strained: the terms pull
        du_dx, du_dp, u = pot.to_gpu(precision).unbound_impl.execute(x, prm, box)
        np.testing.assert_allclose(u, float(g[f"u_{key}"]), rtol=brt, atol=brt * 10)
        assert_equal_vectors(g[f"du_dx_{key}"], du_dx, brt)


@pytest.mark.parametrize("tag", ["vacuum", "pbc"])
def test_config1_md_follows_the_cpu_reference_path(co, P, tag):
    """Config 1 is the reference's CPU-runnable case (SKIP_CUSTOM_OPS: JAX potentials + the Python BAOAB integrator,
    timemachine/integrator.py:124-150).  Friction 0 => deterministic: the GPU Context against the oracle's restatement of
    that path (oracle forces, f64 BAOAB), then a 1 000-step thermostatted run that must stay finite and bound."""
    from oracle import integrator as oi
    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = ts.config1_water_cluster(100.0 if tag == "vacuum" else 3.0)
    N = s.num_atoms
    rng = np.random.default_rng(7)
    v0 = rng.normal(size=(N, 3)) * np.sqrt(0.008314462618 * 300.0 / s.masses)[:, None]
    dt, T = 1.0e-3, 300.0
    bps = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s)]
    ctxt = co.Context(s.coords, v0, s.box, LangevinIntegrator(T, dt, 0.0, s.masses, 1).impl(), bps)
    ca, cb, cc = oi.langevin_coefficients(T, dt, 0.0, s.masses)
    x, v = s.coords.copy(), v0.copy()
    for _ in range(8):
        f = -rp.nonbonded(x, s.nb_params, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)[1]
        f -= rp.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs)[1]
        f -= rp.harmonic_angle(x, s.angle_params, s.box, s.angle_idxs)[1]
        x, v = oi.baoab_step(x, v, f, np.zeros((N, 3)), ca, cb, cc, dt)
        ctxt.step()
        # the bound integrator rounds velocities to float (wrap_kernels.cpp:700): 1e-7 relative on |v| ~ 1 nm/ps per step
        np.testing.assert_allclose(ctxt.get_x_t(), x, rtol=0, atol=2e-6)
        np.testing.assert_allclose(ctxt.get_v_t(), v, rtol=0, atol=5e-4)
    md = co.Context(s.coords, v0, s.box, LangevinIntegrator(T, 2.5e-3 if tag == "pbc" else 1.0e-3, 1.0, s.masses, 2025).impl(),
                    [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)])
    xs, boxes = md.multiple_steps(1000, 250)
    assert xs.shape == (4, N, 3) and np.all(np.isfinite(xs))
    assert np.abs(xs[-1] - s.coords).max() < 5.0  # the droplet evaporates slowly at most; nothing flies off
    ke = 0.5 * np.sum(s.masses[:, None] * md.get_v_t() ** 2)
    assert 0.5 < ke / (1.5 * N * 0.008314462618 * T) < 1.6


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 4: 8 lambda windows on a ~6.3k-atom state, ligand w = lambda * cutoff (4D softcore decoupling)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_config4_lambda_windows(co, P, precision):
    """u_k(x) for the 8 windows of a relative-hydration-shaped state against the reference's energies (config4.npz),
    through execute_batch (one coordinate set x 8 parameter sets = one row of the end-of-run u_kl gather), plus du_dx and
    the ligand's du_dp at one window."""
    from timemachine_amd import testsystems as ts

    g = load("config4.npz")
    lambdas = g["lambdas"]
    systems = [ts.config4_solvated_ligand(float(lam)) for lam in lambdas]
    s0 = systems[0]
    N = s0.num_atoms
    assert 6000 < N < 7000 and s0.box[0, 0] == 4.0 and len(lambdas) == 8
    x = g["x"].astype(np.float64)
    params = np.stack([s.nb_params.astype(np.float32).astype(np.float64) for s in systems])
    lig = np.arange(s0.num_water_atoms, N)
    np.testing.assert_allclose(params[-1][lig, 3], s0.cutoff)  # lambda = 1: the ligand sits at w = cutoff, fully decoupled
    nb = P.Nonbonded(N, s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff).to_gpu(precision).unbound_impl
    _, _, u = nb.execute_batch(x[None], params, s0.box[None], False, False, True)
    t = TOL[precision]
    np.testing.assert_allclose(u[0], g["u_k"], rtol=t["rtol"] * 10, atol=t["atol"] * 10 if precision == np.float64 else 0.5)
    # energy DIFFERENCES between windows are what BAR/MBAR consume: they involve only ligand pairs
    np.testing.assert_allclose(u[0] - u[0][0], g["u_k"] - g["u_k"][0], rtol=0, atol=1e-6 if precision == np.float64 else 0.2)
    k = int(g["grad_state"])
    du_dx, du_dp, uk = nb.execute(x, params[k], s0.box)
    assert uk == u[0][k]
    assert_equal_vectors(g["du_dx"], du_dx, t["rtol"])
    np.testing.assert_allclose(du_dp[lig], g["du_dp_ligand"], rtol=t["prtol"] * 10, atol=t["patol"] * 10)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at its stated size: ~31k atoms, 24 states, neighbour exchange with max_delta_states = 4
# ----------------------------------------------------------------------------------------------------------------
def test_config5_sized_hrex_iteration(co, P):
    """One HREX iteration of the reference's shape (fe/free_energy.py:1148-1200,1537-1551) at >= 30k atoms x 24 states:
    a few replicas run MD, the sparse (replica, state) energy matrix is evaluated with max_delta_states = 4, states are
    exchanged, parameters re-bound.  Checks: matrix entries == the bound potentials' own energies (same kernels, same
    integers); differences between states == the oracle's ligand-pair energies (only ligand pairs change between windows,
    so the oracle is O(ligand x N), not O(N^2)); Newton III exactly; the permutation stays a permutation."""
    from oracle import ref_potentials as rp
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    n_states, max_delta = 24, 4
    lambdas = np.linspace(0.0, 0.46, n_states)
    s0 = ts.config5_complex_sized(0.0)
    N = s0.num_atoms
    assert N >= 30000
    lig = np.arange(s0.num_water_atoms, N)
    params_by_state = np.stack([ts.config5_complex_sized(float(lam)).nb_params for lam in lambdas[:1]] * n_states)
    for k, lam in enumerate(lambdas):
        params_by_state[k][lig, 3] = lam * s0.cutoff
        params_by_state[k][lig, 0] *= 1.0 - 0.5 * lam
    nb_pot = P.Nonbonded(N, s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff)
    unbound = nb_pot.to_gpu(np.float64).unbound_impl
    # three resident replicas (one rank's share of 24 windows over 8 GPUs), each after a short MD run under its own state
    mine = [0, 8, 16]
    coords, ctxts, bound = [], [], []

    def make_context(r):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s0)]
        bps[-1].set_params(params_by_state[r].reshape(-1))
        return co.Context(s0.coords, np.zeros_like(s0.coords), s0.box, LangevinIntegrator(300.0, 0.25e-3, 50.0, s0.masses, 40 + r).impl(), bps), bps

    for r in mine:
        ctxt, bps = make_context(r)
        ctxt.multiple_steps(25, 0)
        ctxts.append(ctxt)
        bound.append(bps[-1])
        coords.append(ctxt.get_x_t())
    # the same three replicas stepped TOGETHER, the way bench.py --mode hrex and an HREX driver step a rank's replicas
    # (hrex.step_replicas -> custom_ops.multiple_steps_group: 31k-atom systems on the listed pipeline, three streams): same bits
    together = [make_context(r)[0] for r in mine]
    hrex.step_replicas(together, 25, group=3)
    for alone_ctxt, grouped_ctxt in zip(ctxts, together):
        np.testing.assert_array_equal(alone_ctxt.get_x_t(), grouped_ctxt.get_x_t())
        np.testing.assert_array_equal(alone_ctxt.get_v_t(), grouped_ctxt.get_v_t())
    del together
    coords = np.stack(coords)
    boxes = np.stack([s0.box] * len(mine))
    dh = hrex.DistributedHREX(n_states, 300.0, max_delta_states=max_delta, world_size=8, rank=0)
    assert dh.local_replicas == mine
    rows = hrex.compute_potential_matrix(unbound, coords, boxes, params_by_state, dh.replica_idx_by_state, max_delta, replicas=mine)
    assert rows.shape == (3, n_states)
    for i, r in enumerate(mine):
        lo, hi = max(0, r - max_delta), min(n_states - 1, r + max_delta)
        assert np.all(np.isfinite(rows[i, lo : hi + 1])) and np.all(np.isinf(np.delete(rows[i], np.arange(lo, hi + 1))))
        # the oracle: only ligand-involving pairs differ between states
        def ligand_energy(state):
            e_env = rp.nonbonded_interaction_group_energy(rp._t(coords[i]), rp._t(params_by_state[state]), rp._t(boxes[i]), lig, None, s0.beta, s0.cutoff)
            e_self = rp.nonbonded_energy(rp._t(coords[i]), rp._t(params_by_state[state]), rp._t(boxes[i]), s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff, atom_idxs=lig)
            return float(e_env) + float(e_self)

        e_r = ligand_energy(r)
        for st in (lo, hi):
            np.testing.assert_allclose(rows[i, st] - rows[i, r], ligand_energy(st) - e_r, rtol=1e-7, atol=1e-5)
    raw = unbound.execute_raw(coords[0], params_by_state[0], s0.box)
    with np.errstate(over="ignore"):
        assert np.all(raw[0].sum(axis=0, dtype=np.uint64) == 0)
    # exchange with a full (synthetic off-rank) matrix: this rank's rows are the measured ones
    U = np.full((n_states, n_states), np.inf)
    diag = np.arange(n_states)
    for r in range(n_states):
        lo, hi = max(0, r - max_delta), min(n_states - 1, r + max_delta)
        U[r, lo : hi + 1] = rows[0, 0] + 0.3 * np.abs(np.arange(lo, hi + 1) - r)
    for i, r in enumerate(mine):
        U[r] = rows[i]
    log_q = -hrex.verify_and_sanitize_potential_matrix(U, dh.replica_idx_by_state) / dh.kT
    pair_idxs, uniforms = hrex.draw_swap_randomness(9, n_states - 1, hrex.get_swap_attempts_per_iter_heuristic(n_states))
    assert len(pair_idxs) == n_states**3
    perm, proposed, accepted = hrex.run_neighbor_swaps(dh.replica_idx_by_state, dh.pairs, log_q, pair_idxs, uniforms)
    assert sorted(perm.tolist()) == diag.tolist() and accepted.sum() > 0 and proposed.sum() == n_states**3
    new_state = np.argsort(perm)
    for i, r in enumerate(mine):
        bound[i].set_params(params_by_state[new_state[r]].reshape(-1))
        ctxts[i].multiple_steps(5, 0)
        assert np.all(np.isfinite(ctxts[i].get_x_t()))


# ----------------------------------------------------------------------------------------------------------------
# the RCCL code path of timemachine_amd.parallel on the one GPU a test box has: a one-rank "nccl" group
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


_NCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from timemachine_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = np.arange(12, dtype=np.float64).reshape(3, 4) + 0.25
full = parallel.gather_rows([0, 1, 2], rows, 3, row_length=4)
assert full.shape == (3, 4) and np.array_equal(full, rows), full
full = parallel.gather_rows([2, 0], rows[:2], 3)  # row length agreed by the extra collective; window 1 stays NaN
assert np.array_equal(full[2], rows[0]) and np.array_equal(full[0], rows[1]) and np.all(np.isnan(full[1])), full
assert parallel.max_over_ranks(3.5) == 3.5 and parallel.sum_over_ranks(1.25) == 1.25
parallel.barrier()
dist.destroy_process_group()
print("NCCL_ONE_RANK_OK")
"""


def test_parallel_collectives_run_over_rccl_on_one_rank():
    """gather_rows / max_over_ranks / sum_over_ranks / barrier through backend "nccl" (= RCCL) with device tensors: what the
    N > 1 bench does between GPUs, as far as a single-GPU box can exercise it (tensor device, dtype and shape handling)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), TM_AMD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_ONE_RANK, repo], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("mode", ["md", "hrex"])
def test_bench_modes_run_with_rccl_collectives_on_one_rank(mode):
    """bench.py end to end with its collectives forced on (one-rank "nccl" group): the barrier / max-over-ranks timing, the
    u_kl gather (md) and the per-frame energy-row gather (hrex) go through RCCL with device tensors; one JSON line comes out."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               TM_AMD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--mode", mode, "--no-cpu-baseline", "--profile-steps", "0", "--equil-scale", "0.1"]
    cmd += ["--steps", "100", "--warmup", "20"] if mode == "md" else ["--steps", "40", "--warmup", "20", "--steps-per-frame", "20", "--windows", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["backend"] == "nccl" and line["n_gpus"] == 1 and line["value"] > 0
    if mode == "md":
        assert line["mbar_gather_ok"] is True


# ----------------------------------------------------------------------------------------------------------------
# tests/test_determinism.py:19-118 -- re-computed frame energies are bitwise identical, and the fixed-point energies of the
# separate bound potentials add up (uint64 wrap-around) to the SummedPotential's, with and without a barostat in the loop
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,rtol,atol", [(np.float64, 1e-8, 1e-8), (np.float32, 1e-4, 1e-6)])
def test_deterministic_energies_of_md_frames(co, P, precision, rtol, atol):
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    groups = [list(range(3 * k, 3 * k + 3)) for k in range((N - 20) // 3)] + [list(range(N - 20, N))]
    bound = ts.bound_potentials(s, precision)
    bps = [bp.to_gpu(precision).bound_impl for bp in bound]
    summed = P.SummedPotential([bp.potential for bp in bound], [bp.params for bp in bound]).to_gpu(precision)
    ref_pot = summed.bind_params_list([bp.params for bp in bound]).bound_impl
    # relax the lattice start first (NVT, strong friction), as the reference minimises before this test
    relax = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), bps)
    relax.multiple_steps(1000, 0)
    x0, v0 = relax.get_x_t(), relax.get_v_t()
    num_steps = 200
    for with_barostat in (False, True):
        movers = None
        if with_barostat:
            baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 25, 1234).impl(bps)
            baro.set_step(0)
            assert baro.get_interval() <= num_steps
            movers = [baro]
        ctxt = co.Context(x0, v0, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 1234).impl(), bps, movers=movers)
        xs, boxes = ctxt.multiple_steps(num_steps, 10)
        assert len(xs) == num_steps // 10
        for x, b in zip(xs, boxes):
            ref_du_dx, ref_U = ref_pot.execute(x, b)
            assert np.all(np.isfinite(ref_du_dx)) and np.linalg.norm(ref_du_dx, axis=1).max() < 25000.0  # minimizer.check_force_norm's bound
            ref_fixed = ref_pot.execute_fixed(x, b)
            test_u, test_fixed = 0.0, np.uint64(0)
            for bp in bps:
                U_fixed = bp.execute_fixed(x, b)
                assert int(U_fixed[0]) != (1 << 63) - 1  # not overflowed
                with np.errstate(over="ignore"):
                    test_fixed = test_fixed + U_fixed[0]
                _, U = bp.execute(x, b)
                test_u += U
                _, U_again = bp.execute(x, b)  # the same frame again: the same bits
                assert U == U_again
                np.testing.assert_array_equal(bp.execute_fixed(x, b), U_fixed)
            assert test_fixed == ref_fixed[0]  # integer energies add exactly
            np.testing.assert_allclose(ref_U, test_u, rtol=rtol, atol=atol)
            again_du_dx, again_U = ref_pot.execute(x, b)
            assert again_U == ref_U
            np.testing.assert_array_equal(again_du_dx, ref_du_dx)


# ----------------------------------------------------------------------------------------------------------------
# tests/test_bonded_stable.py:59-80 and tests/test_harmonic_angle_32bit.py -- the angle term near its singular geometries
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_harmonic_angle_finite_force_with_vanishing_bond_length(co, P, precision):
    """A bond of 1e-9 nm inside an angle: with the stabiliser eps = 1 the forces stay finite (< 1e7), with eps = 0 they blow up
    (the reference marks that case as an expected failure of the same assertion)."""
    angle_idxs = np.array([[0, 1, 2]], dtype=np.int32)
    coords = np.array([(0.0, 0.0, 0.0), (1e-9, 0.0, 0.0), (0.0, 1.0, 0.0)])
    box = np.eye(3) * 100.0
    impl = P.HarmonicAngle(angle_idxs).to_gpu(precision).unbound_impl
    du_dx, _, _ = impl.execute(coords, np.array([(1.0, 1.0, 1.0)]), box, True, False, False)
    assert np.all(np.isfinite(du_dx)) and np.all(np.abs(du_dx) < 1e7)
    du_dx0, _, _ = impl.execute(coords, np.array([(1.0, 1.0, 0.0)]), box, True, False, False)
    assert not (np.all(np.isfinite(du_dx0)) and np.all(np.abs(du_dx0) < 1e7))


def test_linear_triatomic_is_stable_in_f32(co, P):
    """tests/test_harmonic_angle_32bit.py: a nitrile-like H-C-N with an equilibrium angle of pi, 110 000 f32 Langevin steps at
    1 fs: the mean angle stays above 3.0 rad and no coordinate leaves the neighbourhood (the f32 angle force near theta = pi
    is where a naive acos / sin formulation loses everything).  Parameters: typical GAFF-like magnitudes (the reference takes
    them from its force field files at run time)."""
    from timemachine_amd.lib import LangevinIntegrator

    def kahan_angle(a, b, c):  # bonded.py:82-97 with eps = 0
        u, v = a - b, c - b
        nu, nv = np.linalg.norm(u), np.linalg.norm(v)
        return 2 * np.arctan2(np.linalg.norm(nv * u - nu * v), np.linalg.norm(nv * u + nu * v))

    x0 = np.array([(-0.107, 0.0, 0.0), (0.0, 0.0, 0.0), (0.1157, 0.0, 0.0)]) + 5.0  # H, C, N on a line
    masses = np.array([1.008 * 2, 12.011 - 1.008, 14.007])  # hydrogen mass repartitioning as in the reference test
    bonds = P.HarmonicBond(np.array([[0, 1], [1, 2]], dtype=np.int32)).bind(np.array([(310000.0, 0.107), (700000.0, 0.1157)]))
    angle = P.HarmonicAngle(np.array([[0, 1, 2]], dtype=np.int32)).bind(np.array([(400.0, np.pi, 0.0)]))
    bps = [bonds.to_gpu(np.float32).bound_impl, angle.to_gpu(np.float32).bound_impl]
    box = np.eye(3) * 10.0
    ctxt = co.Context(x0, np.zeros_like(x0), box, LangevinIntegrator(300.0, 1.0e-3, 1.0, masses, 2024).impl(), bps)
    ctxt.multiple_steps(10_000, 0)  # burn-in
    xs, _ = ctxt.multiple_steps(100_000, 1000)
    assert len(xs) == 100
    angles = [kahan_angle(x[0], x[1], x[2]) for x in xs]
    assert np.mean(angles) > 3.0, np.mean(angles)
    assert np.amax(np.abs(xs - 5.0)) < 15.0 and np.all(np.isfinite(xs))


def test_two_contexts_driven_from_two_host_threads():
    """Two MD contexts of one device stepped from two Python threads at once: the compiled binding releases the GIL around the call
    and the C ABI's per-device lock keeps the two calls apart (objects are not thread-safe: cpp/src/potential.hpp:7), so both
    trajectories are the ones the contexts take when stepped one after the other, bit for bit."""
    import threading

    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    co.set_device(0)
    s = ts.small_solvated_ligand(lamb=0.3)

    def make(seed):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.rbfe_bound_potentials(s, 20)]
        return co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, seed).impl(), bps)

    ref = []
    for seed in (5, 6):
        c = make(seed)
        c.multiple_steps(300, 0)
        ref.append((c.get_x_t(), c.get_v_t()))
    ctxts = [make(5), make(6)]
    errors = []

    def worker(c):
        try:
            co.set_device(0)
            for _ in range(3):
                c.multiple_steps(100, 0)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(c,)) for c in ctxts]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for c, (x, v) in zip(ctxts, ref):
        np.testing.assert_array_equal(c.get_x_t(), x)
        np.testing.assert_array_equal(c.get_v_t(), v)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_a_fully_excluded_clash_beyond_the_accumulators_range_cancels_whichever_atom_is_the_row(co, P, precision):
    """Two fully excluded atoms 0.11 nm apart with sigma_ij = 0.36 nm push force components of 2e8 ... 6e8 kJ/mol/nm through the tile
    kernel and the exclusion kernel: beyond the fixed-point accumulator's range (2^27 = 1.3e8; the reference's too, k_fixed_point.cuh).
    The two contributions must still cancel to an exact zero, and with either atom as the tile's row: the pair's second atom gets the
    two's-complement negation of the first one's integers, which equals its own conversion only if the slow conversion is odd beyond
    the range as well (csrc/fixed_point.hip.hpp: tm_llrint_odd; the device's llrint saturates asymmetrically, which left +-2^-4
    kJ/mol/nm on the pair depending on the neighbor list's state -- found by scripts/fuzz_campaign.py)."""
    x = np.array([[1.0, 1.0, 1.0], [1.035, 1.03, 1.1], [2.5, 2.5, 2.5], [0.4, 2.1, 0.3]])
    params = np.array([[2.4, 0.18, 1.0, 0.0], [-1.4, 0.18, 1.0, 0.0], [0.3, 0.15, 0.2, 0.0], [-0.3, 0.15, 0.2, 0.0]])
    box = np.eye(3) * 4.0
    excl, scales = np.array([[0, 1]], dtype=np.int32), np.array([[1.0, 1.0]])
    ref = None
    for order in ([0, 1, 2, 3], [1, 0, 2, 3]):
        for hilbert_off in (False, True):
            nb = P.Nonbonded(4, excl, scales, 2.0, 1.2, disable_hilbert_sort=hilbert_off).to_gpu(precision).unbound_impl
            du_dx = nb.execute(x[order], params[order], box, True, False, False)[0]
            parts = nb.get_potentials()
            big = parts[0].execute(x[order], params[order], box, True, False, False)[0]
            assert np.abs(big[:2]).max() > 1.34e8  # (the all-pairs part alone: pinned at the accumulator's range, 2^27)
            back = np.empty_like(du_dx)
            back[order] = du_dx
            if ref is None:
                ref = back
            np.testing.assert_array_equal(back, ref)
    # atoms 0 and 1 feel only atoms 2 and 3 (far, weak): the clash itself is gone to the last bit
    assert np.abs(ref[:2]).max() < 10.0
