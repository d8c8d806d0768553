"""GPU parity tests (run with ``-m gpu`` on an MI355X).  Every call goes Python -> ctypes -> C ABI -> HIP kernels.

Bars (from the reference's own tests):
  * integer / index work (fixed-point conversion, Hilbert permutation, neighbor-list sets, exclusion cancellation,
    repeatability): BIT-EXACT
  * f64 kernels vs the reference potentials: energy rtol 1e-8, forces 1e-8 of the per-atom force norm (floor 1.0), du_dp
    rtol/atol 1e-7 -- the reference's tolerances (tests/nonbonded/test_nonbonded.py:123,161-163); BASELINE's bar is 1e-6
  * f32 kernels: rtol 1e-4 / atol 5e-4 (tests/nonbonded/test_nonbonded.py:194)
Expected values come from tests/golden/*.npz (energies computed by the reference's own Python; see
tests/golden/generate_golden.py) and, for inputs built on the fly, from the oracle/ package.
"""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    assert custom_ops.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return custom_ops


@pytest.fixture(scope="module")
def P():
    from timemachine_amd import potentials

    return potentials


def assert_equal_vectors(truth, test, rtol):
    """OpenMM convention used by the reference (tests/common.py:250-273): error relative to the force norm, floor 1."""
    assert np.all(np.isfinite(truth)) and np.all(np.isfinite(test))
    norms = np.linalg.norm(truth, axis=-1, keepdims=True)
    norms = np.where(norms < 1.0, 1.0, norms)
    err = np.abs((truth - test) / norms)
    assert err.max() <= rtol, f"max relative force error {err.max():.3e} > {rtol:.1e}"
    return err.max()


TOL = {np.float64: dict(rtol=1e-8, atol=1e-8, prtol=1e-7, patol=1e-7), np.float32: dict(rtol=1e-4, atol=5e-4, prtol=1e-3, patol=5e-3)}


def compare_forces(impl, x, params, box, ref_u, ref_du_dx, ref_du_dp, precision):
    """The reference's GradientTest.compare_forces (tests/common.py:275-334): all 8 flag combinations, each executed
    twice and compared bitwise."""
    t = TOL[precision]
    for compute_du_dx, compute_du_dp, compute_u in itertools.product([False, True], repeat=3):
        du_dx, du_dp, u = impl.execute(x, params, box, compute_du_dx, compute_du_dp, compute_u)
        if compute_u:
            np.testing.assert_allclose(u, ref_u, rtol=t["rtol"], atol=t["atol"])
        else:
            assert u is None
        if compute_du_dx:
            assert_equal_vectors(ref_du_dx, du_dx, t["rtol"])
        else:
            assert du_dx is None
        if compute_du_dp:
            np.testing.assert_allclose(du_dp, ref_du_dp, rtol=t["prtol"], atol=t["patol"])
        else:
            assert du_dp is None
        du_dx2, du_dp2, u2 = impl.execute(x, params, box, compute_du_dx, compute_du_dp, compute_u)
        np.testing.assert_array_equal(du_dx, du_dx2)
        np.testing.assert_array_equal(du_dp, du_dp2)
        assert u == u2


# ----------------------------------------------------------------------------------------------------------------
# Nonbonded vs golden vectors
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["nb_small_w0", "nb_small_wrand", "nb_small_whalf"])
def test_nonbonded_golden(co, P, name, precision, nb_path):
    g = load(name + ".npz")
    beta, cutoff = float(g["beta"]), float(g["cutoff"])
    x, p, box = g["x"], g["params"], g["box"]
    N = x.shape[0]
    impl = P.Nonbonded(N, g["exclusion_idxs"], g["scale_factors"], beta, cutoff).to_gpu(precision).unbound_impl
    compare_forces(impl, x, p, box, float(g["u"]), g["du_dx"], g["du_dp"], precision)
    ap = P.NonbondedAllPairs(N, beta, cutoff).to_gpu(precision).unbound_impl
    compare_forces(ap, x, p, box, float(g["u_all_pairs"]), g["du_dx_all_pairs"], g["du_dp_all_pairs"], precision)
    pl = P.NonbondedPairList(g["exclusion_idxs"], g["scale_factors"], beta, cutoff).to_gpu(precision).unbound_impl
    compare_forces(pl, x, p, box, float(g["u_pair_list"]), g["du_dx_pair_list"], g["du_dp_pair_list"], precision)
    ex = P.NonbondedExclusions(g["exclusion_idxs"], g["scale_factors"], beta, cutoff).to_gpu(precision).unbound_impl
    compare_forces(ex, x, p, box, -float(g["u_pair_list"]), -g["du_dx_pair_list"], -g["du_dp_pair_list"], precision)
    sub = P.Nonbonded(N, g["exclusion_idxs"], g["scale_factors"], beta, cutoff, atom_idxs=g["atom_idxs"]).to_gpu(precision).unbound_impl
    compare_forces(sub, x, p, box, float(g["u_subset"]), g["du_dx_subset"], g["du_dp_subset"], precision)
    # atoms outside atom_idxs get exactly zero force (Appendix B.10)
    du_dx, _, _ = sub.execute(x, p, box)
    outside = np.setdiff1d(np.arange(N), g["atom_idxs"])
    assert np.all(du_dx[outside] == 0.0)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("lamb", ["0.0", "0.3", "1.0"])
def test_config2_all_terms(co, P, lamb, precision, nb_path):
    """BASELINE config 2: ~2 300-atom solvated ligand, every term + du/dp vs the reference potentials."""
    from timemachine_amd import testsystems as ts

    g = load(f"config2_lambda{lamb}.npz")
    s = ts.small_solvated_ligand(lamb=float(lamb))
    x, box = g["x"], s.box
    nb = P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(precision).unbound_impl
    compare_forces(nb, x, g["nb_params"], box, float(g["u_nonbonded"]), g["du_dx_nonbonded"], g["du_dp_nonbonded"], precision)
    t = TOL[precision]
    terms = [
        (P.HarmonicBond(s.bond_idxs), s.bond_params, "bond"),
        (P.HarmonicAngle(s.angle_idxs), s.angle_params, "angle"),
        (P.PeriodicTorsion(s.torsion_idxs), s.torsion_params, "torsion"),
    ]
    for pot, prm, key in terms:
        impl = pot.to_gpu(precision).unbound_impl
        du_dx, du_dp, u = impl.execute(x, prm, box)
        # f64: the reference's bonded tolerance (tests/test_bonded.py:29,101,146).  f32: 1e-4 of the force norm.  The fixture's
        # coordinates are strained (generate_golden.strained: every atom moved by N(0, 0.004 nm)), so every bond and angle
        # pulls with ~1e3 kJ/mol/nm and the absolute f32 error of a stiff O-H bond -- k * eps_f32 * r = 4.6e5 * 6e-8 * 0.1 ~
        # 3e-3 kJ/mol/nm, whatever the arithmetic -- is 1e-6 of the norm; a displacement formed in f32 would not pass.
        brt = 1e-7 if precision == np.float64 else 1e-4
        assert np.linalg.norm(g[f"du_dx_{key}"], axis=1).max() > (100.0 if key != "torsion" else 10.0)
        np.testing.assert_allclose(u, float(g[f"u_{key}"]), rtol=brt, atol=brt * 10)
        assert_equal_vectors(g[f"du_dx_{key}"], du_dx, brt)
        np.testing.assert_allclose(du_dp, g[f"du_dp_{key}"], rtol=brt * 10, atol=brt * 100)
    # the whole state as one SummedPotential (how fe.free_energy.get_context packs it)
    pots = [pot for pot, _, _ in terms] + [P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)]
    prms = [prm for _, prm, _ in terms] + [g["nb_params"]]
    summed = P.SummedPotential(pots, prms).to_gpu(precision)
    flat = np.concatenate([np.asarray(q).reshape(-1) for q in prms])
    du_dx, du_dp, u = summed.unbound_impl.execute(x, flat, box)
    ref_u = sum(float(g[f"u_{k}"]) for k in ("bond", "angle", "torsion", "nonbonded"))
    ref_dx = sum(g[f"du_dx_{k}"] for k in ("bond", "angle", "torsion", "nonbonded"))
    np.testing.assert_allclose(u, ref_u, rtol=t["rtol"] * 10, atol=t["atol"] * 10)
    assert_equal_vectors(ref_dx, du_dx, t["rtol"] * 10)
    assert du_dp.shape == flat.shape
    np.testing.assert_allclose(du_dp[-s.num_atoms * 4 :].reshape(-1, 4), g["du_dp_nonbonded"], rtol=t["prtol"], atol=t["patol"])
    # serial children give the same bits as parallel children (integer accumulation)
    serial = P.SummedPotential(pots, prms, parallel=False).to_gpu(precision)
    du_dx_s, du_dp_s, u_s = serial.unbound_impl.execute(x, flat, box)
    np.testing.assert_array_equal(du_dx, du_dx_s)
    np.testing.assert_array_equal(du_dp, du_dp_s)
    assert u == u_s
    # forces only (the MD path) runs bonded terms + exclusions in ONE fused launch: same bits again, and the same bits
    # as the fixed-point sum of the children executed one by one
    du_dx_f, none_p, none_u = serial.unbound_impl.execute(x, flat, box, True, False, False)
    assert none_p is None and none_u is None
    np.testing.assert_array_equal(du_dx_f, du_dx)
    acc = np.zeros(x.size, dtype=np.uint64)
    for pot, prm in zip(pots, prms):
        acc += pot.to_gpu(precision).unbound_impl.execute_raw(x, np.asarray(prm, dtype=np.float64), box, True, False, False)[0].reshape(-1)
    np.testing.assert_array_equal(acc.view(np.int64).astype(np.float64).reshape(x.shape) / co.FIXED_EXPONENT, du_dx_f)


# ----------------------------------------------------------------------------------------------------------------
# Interaction groups, precomputed pair lists, chiral restraints vs the reference's goldens
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_interaction_group_golden(co, P, precision):
    """reference tests: tests/nonbonded/test_nonbonded_interaction_group.py (vs nonbonded_interaction_groups, and the
    all-pairs decomposition of test_consistency.py)."""
    g = load("groups.npz")
    x, p, box, rows = g["ig_x"], g["ig_params"], g["ig_box"], g["ig_rows"]
    beta, cutoff = float(g["beta"]), float(g["cutoff"])
    n = x.shape[0]
    for tag, cols in (("ig_all", None), ("ig_sub", g["ig_cols_sub"])):
        pot = P.NonbondedInteractionGroup(n, rows, beta, cutoff, col_atom_idxs=cols)
        impl = pot.to_gpu(precision).unbound_impl
        compare_forces(impl, x, p, box, float(g[f"{tag}_u"]), g[f"{tag}_du_dx"], g[f"{tag}_du_dp"], precision)
        # Hilbert sorting off / a different padding: same bits
        ref = impl.execute_raw(x, p, box)
        for kw in (dict(disable_hilbert_sort=True), dict(nblist_padding=0.25)):
            other = P.NonbondedInteractionGroup(n, rows, beta, cutoff, col_atom_idxs=cols, **kw).to_gpu(precision).unbound_impl
            got = other.execute_raw(x, p, box)
            np.testing.assert_array_equal(ref[0], got[0])
            np.testing.assert_array_equal(ref[1], got[1])
            assert ref[2] == got[2]
    # bitwise decomposition in fixed point: AllPairs(rows U cols) == AllPairs(rows) + AllPairs(cols) + Group(rows, cols)
    cols = g["ig_cols_sub"]
    union = np.sort(np.concatenate([rows, cols])).astype(np.int32)
    parts = [
        P.NonbondedAllPairs(n, beta, cutoff, atom_idxs=rows.astype(np.int32)),
        P.NonbondedAllPairs(n, beta, cutoff, atom_idxs=cols.astype(np.int32)),
        P.NonbondedInteractionGroup(n, rows, beta, cutoff, col_atom_idxs=cols),
    ]
    whole = P.NonbondedAllPairs(n, beta, cutoff, atom_idxs=union).to_gpu(precision).unbound_impl.execute_raw(x, p, box)
    acc_x, acc_p, acc_u = np.zeros_like(whole[0]), np.zeros_like(whole[1]), 0
    for part in parts:
        gx, gp, u = part.to_gpu(precision).unbound_impl.execute_raw(x, p, box)
        acc_x += gx
        acc_p += gp
        acc_u += u
    np.testing.assert_array_equal(acc_x, whole[0])
    np.testing.assert_array_equal(acc_p, whole[1])
    assert acc_u == whole[2]
    # set_atom_idxs swaps the groups in place
    impl = P.NonbondedInteractionGroup(n, rows, beta, cutoff).to_gpu(precision).unbound_impl
    impl.set_atom_idxs(rows, cols)
    du_dx, _, u = impl.execute(x, p, box, True, False, True)
    t = TOL[precision]
    np.testing.assert_allclose(u, float(g["ig_sub_u"]), rtol=t["rtol"], atol=t["atol"])
    assert_equal_vectors(g["ig_sub_du_dx"], du_dx, t["rtol"])
    # validation (nonbonded_interaction_group.cu:353-381)
    ctor = co.NonbondedInteractionGroup_f32
    with pytest.raises(RuntimeError, match="row_atom_idxs must be nonempty"):
        ctor(n, np.zeros(0, np.int32), beta, cutoff)
    with pytest.raises(RuntimeError, match="row and col indices must be disjoint"):
        ctor(n, np.array([0, 1], np.int32), beta, cutoff, np.array([1, 2], np.int32))
    with pytest.raises(RuntimeError, match="atom indices must be unique"):
        ctor(n, np.array([0, 0], np.int32), beta, cutoff)
    with pytest.raises(RuntimeError, match=r"NonbondedInteractionGroup::execute_device\(\): expected N == N_"):
        ctor(n, np.array([0], np.int32), beta, cutoff).execute(x[:-1], p[:-1], box)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_pair_list_precomputed_golden(co, P, precision):
    """reference test: tests/nonbonded/test_nonbonded_precomputed.py"""
    g = load("groups.npz")
    impl = P.NonbondedPairListPrecomputed(g["pre_idxs"], float(g["beta"]), float(g["cutoff"])).to_gpu(precision).unbound_impl
    compare_forces(impl, g["ig_x"], g["pre_params"], g["ig_box"], float(g["pre_u"]), g["pre_du_dx"], g["pre_du_dp"], precision)
    with pytest.raises(RuntimeError, match="illegal pair with src == dst: 3, 3"):
        co.NonbondedPairListPrecomputed_f32(np.array([[3, 3]], np.int32), 2.0, 1.2)
    with pytest.raises(RuntimeError, match=r"expected P == 4\*B"):
        impl.execute(g["ig_x"], g["pre_params"][:-1], g["ig_box"])


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_chiral_restraints_golden(co, P, precision):
    """reference tests: tests/test_chiral_restraints.py (GPU vs chiral_atom_restraint / chiral_bond_restraint)"""
    g = load("groups.npz")
    x = g["chiral_x"]
    box = np.eye(3) * 100.0
    brt = 1e-7 if precision == np.float64 else 2e-3
    for key, pot, prm in (
        ("chiral_atom", P.ChiralAtomRestraint(g["chiral_atom_idxs"]), g["chiral_atom_params"]),
        ("chiral_bond", P.ChiralBondRestraint(g["chiral_bond_idxs"], g["chiral_bond_signs"]), g["chiral_bond_params"]),
    ):
        impl = pot.to_gpu(precision).unbound_impl
        for flags in itertools.product([False, True], repeat=3):
            du_dx, du_dp, u = impl.execute(x, prm, box, *flags)
            if flags[2]:
                np.testing.assert_allclose(u, float(g[f"{key}_u"]), rtol=brt, atol=brt * 10)
            if flags[0]:
                assert_equal_vectors(g[f"{key}_du_dx"], du_dx, brt)
            if flags[1]:
                np.testing.assert_allclose(du_dp, g[f"{key}_du_dp"], rtol=brt * 10, atol=brt * 10)
            again = impl.execute(x, prm, box, *flags)
            for a, b in zip((du_dx, du_dp), again[:2]):
                np.testing.assert_array_equal(a, b)
            assert u == again[2]
    with pytest.raises(RuntimeError, match="signs must be comprised exclusively of 1 or -1"):
        co.ChiralBondRestraint_f32(np.array([[0, 1, 2, 3]], np.int32), np.array([0], np.int32))
    with pytest.raises(RuntimeError, match=r"ChiralAtomRestraint::execute_device\(\): expected P == R"):
        P.ChiralAtomRestraint(g["chiral_atom_idxs"]).to_gpu(precision).unbound_impl.execute(x, g["chiral_atom_params"][:-1], box)


@pytest.mark.gpu
def test_hrex_energy_matrix_and_exchange_on_one_gpu(co, P):
    """BASELINE config 5 shape on one GPU: 6 lambda windows of the solvated-ligand state.  The sparse (replica, state)
    energy matrix equals the dense execute_batch wherever it is evaluated (fe/free_energy.py:1148-1200); an exchange step
    re-assigns states, and a replica's Context then runs under its new parameters (states move, coordinates do not)."""
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    n_states = 6
    lambdas = np.linspace(0.0, 0.5, n_states)
    systems = [ts.small_solvated_ligand(lamb=float(lam)) for lam in lambdas]
    s0 = systems[0]
    params_by_state = np.stack([s.nb_params for s in systems])  # windows differ only in the ligand's nonbonded params
    assert not np.array_equal(params_by_state[0], params_by_state[-1])
    nb = P.Nonbonded(s0.num_atoms, s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff)
    unbound = nb.to_gpu(np.float32).unbound_impl
    # one short trajectory per replica so that the replicas differ
    ctxts, bound_nb = [], []
    for k in range(n_states):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(systems[k])]
        bound_nb.append(bps[-1])
        ctxt = co.Context(s0.coords, np.zeros_like(s0.coords), s0.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s0.masses, 20 + k).impl(), bps)
        ctxt.multiple_steps(30, 0)
        ctxts.append(ctxt)
    coords = np.stack([c.get_x_t() for c in ctxts])
    boxes = np.stack([c.get_box() for c in ctxts])
    dh = hrex.DistributedHREX(n_states, 300.0, max_delta_states=1)
    rows = hrex.compute_potential_matrix(unbound, coords, boxes, params_by_state, dh.replica_idx_by_state, dh.max_delta_states)
    _, _, dense = unbound.execute_batch(coords, params_by_state, boxes, False, False, True)
    evaluated = np.isfinite(rows)
    assert evaluated.sum() == 3 * n_states - 2 and np.all(evaluated[np.arange(n_states), np.arange(n_states)])
    np.testing.assert_array_equal(rows[evaluated], dense[evaluated])  # same kernels, same integer sums
    # a rank's share of the rows (replicas 1, 3, 5 of a 2-rank job) is the same numbers
    mine = [1, 3, 5]
    part = hrex.compute_potential_matrix(unbound, coords[mine], boxes[mine], params_by_state, dh.replica_idx_by_state, 1, replicas=mine)
    np.testing.assert_array_equal(part, rows[mine])
    # exchange, then every replica adopts the parameters of its new state
    new_states = dh.exchange(rows, seed=5)
    assert sorted(new_states.tolist()) == list(range(n_states))
    for r in range(n_states):
        bound_nb[r].set_params(params_by_state[new_states[r]].reshape(-1))
        _, u = bound_nb[r].execute(coords[r], boxes[r], False, True)
        assert u == dense[r, new_states[r]]
        ctxts[r].multiple_steps(5, 0)
        assert np.all(np.isfinite(ctxts[r].get_x_t()))


@pytest.mark.gpu
def test_barostat_follows_model_attempt_by_attempt(co, P):
    """MonteCarloBarostat: every attempt's proposal (molecular centroid scaling) and Metropolis
    decision against oracle/barostat.py, fed with the same Philox uniforms and the GPU's own energies.
    reference tests: tests/test_barostat.py (scaling keeps molecules rigid, interval / validation semantics)."""
    from oracle import barostat as ob
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import MonteCarloBarostat

    s = _md_system()  # waters first, then the 16-atom chain ligand
    N = s.num_atoms
    groups = [list(range(3 * k, 3 * k + 3)) for k in range((N - 16) // 3)] + [list(range(N - 16, N))]
    assert sum(len(g) for g in groups) == N
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    eval_bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    seed, T, pressure = 2024, 300.0, 1.0
    baro = MonteCarloBarostat(N, pressure, T, groups, 1, seed, adaptive_scaling_enabled=False, initial_volume_scale_factor=0.0).impl(bps)
    baro.set_volume_scale_factor(0.35)  # nm^3: large enough that some attempts are rejected
    assert baro.get_volume_scale_factor() == 0.35 and baro.get_adaptive_scaling() is False and baro.get_interval() == 1

    def energy(x, box):
        return sum(bp.execute(x, box, False, True)[1] for bp in eval_bps)

    x, box = s.coords.copy(), s.box.copy()
    n_accept = 0
    for attempt in range(8):
        u1, u2 = ob.attempt_uniforms(seed, attempt)
        x_prop, box_prop, (vol, delta, scale) = ob.propose(x, box, groups, 0.35, u1)
        ok, w = ob.accept(energy(x, box), energy(x_prop, box_prop), vol, delta, len(groups), T, pressure, u2)
        x_new, box_new = baro.move(x, box)
        if abs(w) > 1e-2 * 2.5:  # decisions within rounding distance of the threshold are not compared
            assert ok == (not np.array_equal(box_new, box)), (attempt, ok, w)
        if not np.array_equal(box_new, box):
            n_accept += 1
            np.testing.assert_allclose(box_new, box_prop, rtol=1e-6)
            np.testing.assert_allclose(x_new, x_prop, rtol=0, atol=2e-5)
            for g in groups[:50] + groups[-1:]:  # molecules move rigidly
                d_old = x[g][:, None, :] - x[g][None, :, :]
                d_new = x_new[g][:, None, :] - x_new[g][None, :, :]
                np.testing.assert_allclose(d_new, d_old, rtol=0, atol=1e-12)
            cent = np.array([x_new[g].mean(0) for g in groups])
            assert np.all(cent >= -1e-5) and np.all(cent <= np.diagonal(box_new) + 1e-5)  # centroids in the home box
        else:
            np.testing.assert_array_equal(x_new, x)
        x, box = x_new, box_new
    assert 0 < n_accept < 8, n_accept
    assert sum(baro.get_counters()) > 0

    # interval semantics (mover.hpp:23-40): acts on every 3rd call counted from set_interval
    baro.set_interval(3)
    moved = []
    for _ in range(6):
        x_new, box_new = baro.move(x, box)
        moved.append(not np.array_equal(x_new, x) or not np.array_equal(box_new, box))
    assert moved[0] is False and moved[1] is False and moved[3] is False and moved[4] is False
    with pytest.raises(RuntimeError, match="interval must be greater than 0"):
        baro.set_interval(0)
    with pytest.raises(RuntimeError, match="step must be at least 0"):
        baro.set_step(-1)
    with pytest.raises(RuntimeError, match="All grouped indices must be unique"):
        MonteCarloBarostat(N, 1.0, T, [[0, 1], [1, 2]], 1, 1).impl(bps)
    with pytest.raises(RuntimeError, match="Grouped indices must be between 0 and N"):
        MonteCarloBarostat(N, 1.0, T, [[0, N]], 1, 1).impl(bps)
    # adaptive scaling: 0 means "1 % of the volume" on the first attempt
    adaptive = MonteCarloBarostat(N, pressure, T, groups, 1, seed).impl(bps)
    assert adaptive.get_adaptive_scaling() is True and adaptive.get_volume_scale_factor() == 0.0
    adaptive.move(x, box)
    np.testing.assert_allclose(adaptive.get_volume_scale_factor(), 0.01 * np.prod(np.diagonal(box)), rtol=1e-6)


@pytest.mark.gpu
def test_npt_context_runs_with_barostat(co, P):
    """Context with a barostat mover (fe/free_energy.py:695-708 shape): box fluctuates, density stays liquid-like,
    get_barostat() finds the mover, adaptive scaling settles between 25 % and 75 % acceptance."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand()  # liquid density: ~770 waters + a 20-atom ligand in 2.85 nm
    N = s.num_atoms
    groups = [list(range(3 * k, 3 * k + 3)) for k in range((N - 20) // 3)] + [list(range(N - 20, N))]
    assert sum(len(g) for g in groups) == N
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 5, 7).impl(bps)
    # relax the synthetic lattice start at constant volume first (it is far from equilibrium: the barostat would
    # spend the whole test expanding the box)
    nvt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), bps)
    nvt.multiple_steps(1500, 0)
    ctxt = co.Context(nvt.get_x_t(), nvt.get_v_t(), s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 3).impl(), bps, movers=[baro])
    assert ctxt.get_barostat() is baro and ctxt.get_movers() == [baro]
    xs, boxes = ctxt.multiple_steps(600, 100)
    vols = np.prod(np.diagonal(boxes, axis1=1, axis2=2), axis=1)
    v0 = np.prod(np.diagonal(s.box))
    assert np.all(np.isfinite(xs)) and len(np.unique(vols)) > 1
    assert np.all(np.abs(vols / v0 - 1) < 0.25), vols / v0  # a sanity bound, not an equation of state
    assert baro.get_volume_scale_factor() > 0


@pytest.mark.gpu
def test_velocity_verlet_matches_device_model_and_reverses(co, P):
    """VelocityVerletIntegrator: multiple_steps == initialize + (n-1) steps + finalize of the device
    model with the GPU's own fixed-point forces; time reversal returns to the start (tests/test_velocity_verlet_integrator.py)."""
    from oracle import integrator as oi
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import VelocityVerletIntegrator

    s = _md_system()
    N = s.num_atoms
    rng = np.random.default_rng(5)
    v0 = rng.normal(size=(N, 3)) * 0.2
    dt, n_steps = 0.5e-3, 10
    vv = VelocityVerletIntegrator(dt, s.masses)
    np.testing.assert_array_equal(vv.cbs, -dt / s.masses)
    bps = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s)]
    eval_bps = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s)]

    def force_fixed(x):
        fixed = np.zeros((N, 3), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for bp in eval_bps:
                dx, _ = bp.execute(x, s.box, True, False)
                fixed += np.rint(dx * 2.0**36).astype(np.int64).view(np.uint64)
        return fixed

    ctxt = co.Context(s.coords, v0, s.box, vv.impl(), bps)
    xs, _ = ctxt.multiple_steps(n_steps, 0)
    # Context.multiple_steps(n) = initialize + n steps + finalize, i.e. n + 1 drifts: the reference's own test compares it
    # with the Python integrator's multiple_steps(n + 1) (tests/test_velocity_verlet_integrator.py:128-136)
    x_model, v_model = oi.velocity_verlet_device_model(s.coords, v0, force_fixed, vv.cbs, dt, n_steps + 1)
    np.testing.assert_allclose(ctxt.get_x_t(), x_model, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ctxt.get_v_t(), v_model, rtol=0, atol=1e-10)
    np.testing.assert_array_equal(xs[-1], ctxt.get_x_t())
    # reversibility: flip the velocities, integrate the same number of steps, land on the start
    x1, v1 = ctxt.get_x_t(), ctxt.get_v_t()
    back = co.Context(x1, -v1, s.box, vv.impl(), bps)
    back.multiple_steps(n_steps, 0)
    np.testing.assert_allclose(back.get_x_t(), s.coords, rtol=0, atol=1e-9)
    np.testing.assert_allclose(back.get_v_t(), -v0, rtol=0, atol=1e-7)
    # initialize / finalize bookkeeping (verlet_integrator.cu:52-54,88-90)
    c2 = co.Context(s.coords, v0, s.box, vv.impl(), bps)
    c2.initialize()
    with pytest.raises(RuntimeError, match="initialized twice"):
        c2.initialize()
    c2.finalize()
    with pytest.raises(RuntimeError, match="not initialized"):
        c2.finalize()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_flat_bottom_and_centroid_restraints_golden(co, P, precision):
    """reference tests: tests/test_bonded.py (flat_bottom_bond, log_flat_bottom_bond), tests/test_centroid_restraint.py"""
    g = load("groups.npz")
    x, box = g["chiral_x"], g["fb_box"]
    brt = 1e-7 if precision == np.float64 else 2e-3
    cases = (
        ("fb", P.FlatBottomBond(g["fb_idxs"]), g["fb_params"]),
        ("lfb", P.LogFlatBottomBond(g["lfb_idxs"], float(g["lfb_beta"])), g["lfb_params"]),
    )
    for key, pot, prm in cases:
        impl = pot.to_gpu(precision).unbound_impl
        for flags in itertools.product([False, True], repeat=3):
            du_dx, du_dp, u = impl.execute(x, prm, box, *flags)
            if flags[2]:
                np.testing.assert_allclose(u, float(g[f"{key}_u"]), rtol=brt, atol=brt * 10)
            if flags[0]:
                assert_equal_vectors(g[f"{key}_du_dx"], du_dx, brt)
            if flags[1]:
                np.testing.assert_allclose(du_dp, g[f"{key}_du_dp"], rtol=brt * 10, atol=brt * 100)
            again = impl.execute(x, prm, box, *flags)
            for a, b in zip((du_dx, du_dp), again[:2]):
                np.testing.assert_array_equal(a, b)
            assert u == again[2]
    for tag in ("cr", "cr0"):
        impl = P.CentroidRestraint(g["cr_a"], g["cr_b"], float(g[f"{tag}_kb"]), float(g[f"{tag}_b0"])).to_gpu(precision).unbound_impl
        du_dx, _, u = impl.execute(x, np.zeros(0), np.eye(3) * 100.0, True, False, True)
        np.testing.assert_allclose(u, float(g[f"{tag}_u"]), rtol=brt, atol=brt * 10)
        assert_equal_vectors(g[f"{tag}_du_dx"], du_dx, brt)
    with pytest.raises(RuntimeError, match="beta must be positive"):
        co.LogFlatBottomBond_f32(np.array([[0, 1]], np.int32), 0.0)
    with pytest.raises(RuntimeError, match=r"FlatBottomBond::execute_device\(\): expected P == 3\*B"):
        P.FlatBottomBond(g["fb_idxs"]).to_gpu(precision).unbound_impl.execute(x, g["fb_params"][:-1], box)


@pytest.mark.gpu
def test_rbfe_shaped_state_runs_fused(co, P):
    """A HostGuestSystem-shaped state (fe/system.py:132-143): host-host Nonbonded(atom_idxs=host), ligand-environment
    interaction group, ligand-ligand precomputed pairs, chiral restraints, bonded terms -- as one SummedPotential in an
    MD Context.  Forces-only evaluation (one fused launch + two tile launches) must equal the sum of the parts bit for bit."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = ts.small_solvated_ligand(lamb=0.3)
    n = s.num_atoms
    lig = np.arange(n - 20, n, dtype=np.int32)
    host = np.arange(0, n - 20, dtype=np.int32)
    rng = np.random.default_rng(7)
    lig_pairs = np.array([(i, j) for i in lig for j in lig if i < j], dtype=np.int32)[::3]
    pre_params = np.stack([rng.normal(size=len(lig_pairs)), rng.uniform(0.2, 0.3, len(lig_pairs)), rng.uniform(0.1, 0.5, len(lig_pairs)), np.zeros(len(lig_pairs))], 1)
    chiral_idxs = np.stack([rng.permutation(lig)[:4] for _ in range(6)]).astype(np.int32)
    pots = [
        (P.HarmonicBond(s.bond_idxs), s.bond_params),
        (P.HarmonicAngle(s.angle_idxs), s.angle_params),
        (P.PeriodicTorsion(s.torsion_idxs), s.torsion_params),
        (P.Nonbonded(n, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, atom_idxs=host), s.nb_params),
        (P.NonbondedInteractionGroup(n, lig, s.beta, s.cutoff), s.nb_params),
        (P.NonbondedPairListPrecomputed(lig_pairs, s.beta, s.cutoff), pre_params),
        (P.ChiralAtomRestraint(chiral_idxs), np.full(6, 100.0)),
        (P.ChiralBondRestraint(chiral_idxs, np.array([1, -1, 1, -1, 1, -1], np.int32)), np.full(6, 100.0)),
    ]
    x, box = s.coords, s.box
    summed = P.SummedPotential([p for p, _ in pots], [q for _, q in pots], parallel=False).to_gpu(np.float32).unbound_impl
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in pots])
    fused = summed.execute_raw(x, flat, box, True, False, False)[0]
    acc = np.zeros_like(fused)
    for pot, prm in pots:
        acc += pot.to_gpu(np.float32).unbound_impl.execute_raw(x, np.asarray(prm, dtype=np.float64), box, True, False, False)[0]
    np.testing.assert_array_equal(acc, fused)
    # and it integrates
    bps = [pot.bind(prm).to_gpu(np.float32).bound_impl for pot, prm in pots]
    ctxt = co.Context(x, np.zeros_like(x), box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 11).impl(), bps)
    xs, _ = ctxt.multiple_steps(50, 50)
    assert np.all(np.isfinite(xs))
    assert np.abs(xs[-1] - x).max() < 0.5


# ----------------------------------------------------------------------------------------------------------------
# Bitwise invariances (integer accumulation => reordering must not change one bit)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_bitwise_invariances(co, P, precision):
    from timemachine_amd import testsystems as ts

    s = ts.small_solvated_ligand(lamb=0.3)
    x, p, box = s.coords, s.nb_params, s.box
    N = s.num_atoms

    def raw(pot, xx=x, pp=p):
        return pot.to_gpu(precision).unbound_impl.execute_raw(xx, pp, box)

    base = raw(P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff))
    # Hilbert sort on/off (tests/nonbonded/test_nonbonded.py:18-60)
    nohilb = raw(P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, disable_hilbert_sort=True))
    # padding 0.0 vs 0.1 (test_nblist_rebuild)
    nopad = raw(P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, nblist_padding=0.0))
    # atom_idxs=None vs arange(N)
    arange = raw(P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, atom_idxs=np.arange(N, dtype=np.int32)))
    for other in (nohilb, nopad, arange):
        np.testing.assert_array_equal(base[0], other[0])
        np.testing.assert_array_equal(base[1], other[1])
        assert base[2] == other[2]
    # random relabelling of the atoms permutes the output, bit for bit
    rng = np.random.default_rng(5)
    perm = rng.permutation(N)
    inv = np.argsort(perm)
    excl_perm = inv[s.exclusion_idxs].astype(np.int32)
    permuted = raw(P.Nonbonded(N, excl_perm, s.scale_factors, s.beta, s.cutoff), x[perm], p[perm])
    np.testing.assert_array_equal(base[0][perm], permuted[0])
    np.testing.assert_array_equal(base[1].reshape(N, 4)[perm].reshape(-1), permuted[1])
    assert base[2] == permuted[2]
    # Newton's third law in fixed point: sum of all force accumulators wraps to exactly zero
    with np.errstate(over="ignore"):
        assert np.all(base[0].sum(axis=0, dtype=np.uint64) == 0)
    # decomposition: AllPairs + Exclusions == Nonbonded, bitwise (tests/nonbonded/test_consistency.py)
    ap = raw(P.NonbondedAllPairs(N, s.beta, s.cutoff))
    ex = raw(P.NonbondedExclusions(s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff))
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(ap[0] + ex[0], base[0])
        np.testing.assert_array_equal(ap[1] + ex[1], base[1])
    assert ap[2] + ex[2] == base[2]


def test_large_box_decomposes_bitwise(co, P):
    """69 696 atoms, cutoff 1.5 nm: more than 2048 column blocks (the list build walks them in two LDS chunks) and row
    blocks that list more column atoms than the build stages in LDS (the cost estimate's global fallback).  The
    all-pairs result must equal, bit for bit, the sum of the two halves and their interaction group -- which stay on the
    single-chunk paths (the decomposition of the reference's tests/nonbonded/test_consistency.py, at a size that reaches
    the code the small systems do not)."""
    from timemachine_amd import testsystems as ts

    s = ts.build_water_box(23232, 8.86, seed=9)
    N = s.num_atoms
    assert N > 65536
    cutoff = 1.5
    x, p, box = s.coords, s.nb_params, s.box
    a_idxs = np.arange(0, (N // 6) * 3, dtype=np.int32)
    b_idxs = np.arange((N // 6) * 3, N, dtype=np.int32)

    def raw(pot):
        return pot.to_gpu(np.float32).unbound_impl.execute_raw(x, p, box)

    whole = raw(P.NonbondedAllPairs(N, s.beta, cutoff))
    parts = [
        raw(P.NonbondedAllPairs(N, s.beta, cutoff, atom_idxs=a_idxs)),
        raw(P.NonbondedAllPairs(N, s.beta, cutoff, atom_idxs=b_idxs)),
        raw(P.NonbondedInteractionGroup(N, a_idxs, s.beta, cutoff, col_atom_idxs=b_idxs)),
    ]
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(parts[0][0] + parts[1][0] + parts[2][0], whole[0])
        np.testing.assert_array_equal(parts[0][1] + parts[1][1] + parts[2][1], whole[1])
        assert np.all(whole[0].sum(axis=0, dtype=np.uint64) == 0)
    assert parts[0][2] + parts[1][2] + parts[2][2] == whole[2]
    assert whole[2] != 0


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_exclusions_cancel_exactly_with_clashing_atoms(co, P, precision):
    """10 mutually excluded atoms within 1e-3 nm of each other: the all-pairs terms are astronomically large and must
    cancel bit-for-bit against the exclusion kernel (tests/nonbonded/test_nonbonded.py:208-251)."""
    from oracle import ref_potentials as rp

    rng = np.random.default_rng(11)
    N, L, beta, cutoff = 96, 3.0, 2.0, 1.2
    x = rng.uniform(0, L, (N, 3))
    x[:10] = x[0] + rng.uniform(-1e-3, 1e-3, (10, 3))
    params = np.stack([(rng.uniform(size=N) - 0.5) * 11.0, rng.uniform(0.05, 0.1, N), np.sqrt(rng.uniform(size=N)), np.zeros(N)], 1)
    x = x.astype(np.float32).astype(np.float64)
    params = params.astype(np.float32).astype(np.float64)
    excl = np.array([(i, j) for i in range(10) for j in range(i + 1, 10)], dtype=np.int32)
    scales = np.ones((len(excl), 2))
    box = np.eye(3) * L
    ref_u, ref_dx, ref_dp = rp.nonbonded(x, params, box, excl, scales, beta, cutoff)
    impl = P.Nonbonded(N, excl, scales, beta, cutoff).to_gpu(precision).unbound_impl
    du_dx, du_dp, u = impl.execute(x, params, box)
    t = TOL[precision]
    np.testing.assert_allclose(u, ref_u, rtol=t["rtol"], atol=t["atol"])
    assert_equal_vectors(ref_dx, du_dx, t["rtol"])
    np.testing.assert_allclose(du_dp, ref_dp, rtol=t["prtol"], atol=t["patol"])
    # without the exclusions the same configuration must overflow the energy (and only the energy) -> NaN
    ap = P.NonbondedAllPairs(N, beta, cutoff).to_gpu(precision).unbound_impl
    _, _, u_ap = ap.execute(x, params, box, False, False, True)
    assert np.isnan(u_ap) or abs(u_ap) > 1e6


def test_energy_overflow_semantics(co, P):
    """Overlapping atoms: NaN from execute, LLONG_MAX from execute_fixed (tests/test_energy_overflows.py:21-63)."""
    N, L = 8, 4.0
    x = np.zeros((N, 3)) + np.arange(N)[:, None] * 0.3
    x[1] = x[0] + 1e-12
    params = np.tile([3.0, 0.15, 1.0, 0.0], (N, 1))
    box = np.eye(3) * L
    for precision in (np.float32, np.float64):
        ap = P.NonbondedAllPairs(N, 2.0, 1.2).to_gpu(precision)
        _, _, u = ap.unbound_impl.execute(x, params, box, False, False, True)
        assert np.isnan(u)
        fixed = ap.bind(params).bound_impl.execute_fixed(x, box)
        assert fixed.dtype == np.uint64 and int(fixed[0]) == (1 << 63) - 1


# ----------------------------------------------------------------------------------------------------------------
# Bonded
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_bonded_golden_and_symmetry(co, P, precision):
    g = load("bonded.npz")
    x, box = g["x"], g["box"]
    # f32: 1e-5 of the force norm (measured 6.2e-6: bonds of this fixture are up to 1.5 nm long, and k * eps_f32 * r is what an
    # f32 bond force is uncertain by however the arithmetic is arranged; 2e-5 before the displacements were formed in double)
    rt = 1e-7 if precision == np.float64 else 1e-5
    for cls, key, width in ((P.HarmonicBond, "bond", 2), (P.HarmonicAngle, "angle", 3), (P.PeriodicTorsion, "torsion", 4)):
        idxs, prm = g[f"{key}_idxs"], g[f"{key}_params"]
        impl = cls(idxs).to_gpu(precision).unbound_impl
        # the same atoms 50 nm from the origin (x_far = x + 50 exactly): bonded terms see differences only, and the
        # reference forms them in double before casting to the kernel's precision (k_harmonic_bond.cuh:27,
        # k_harmonic_angle.cuh:44-45, k_periodic_torsion.cuh:49-51) -- so must every f32 kernel here: same tolerance, and the
        # very same bits as at the origin (a subtraction after the cast would be off by 4e-6 nm * k)
        far = impl.execute_raw(g["x_far"], prm, box, True, False, True)
        near = impl.execute_raw(x, prm, box, True, False, True)
        np.testing.assert_array_equal(far[0], near[0])
        assert far[2] == near[2]
        du_dx_far, _, u_far = impl.execute(g["x_far"], prm, box, True, False, True)
        assert_equal_vectors(g[f"du_dx_{key}"], du_dx_far, rt)
        np.testing.assert_allclose(u_far, float(g[f"u_{key}_far"]), rtol=rt, atol=rt * 10)
        for cx, cp, cu in itertools.product([False, True], repeat=3):
            du_dx, du_dp, u = impl.execute(x, prm, box, cx, cp, cu)
            if cu:
                np.testing.assert_allclose(u, float(g[f"u_{key}"]), rtol=rt, atol=rt * 10)
            if cx:
                assert_equal_vectors(g[f"du_dx_{key}"], du_dx, rt)
            if cp:
                ref_dp = g[f"du_dp_{key}"].copy()
                if key == "bond":
                    # b0 == 0 rows: the reference's CUDA kernel reports du/db0 = -k (r - b0) unconditionally
                    # (k_harmonic_bond.cuh:49-52) while its JAX energy switches to k/2 r^2 there (bonded.py:74-77), whose
                    # b0-derivative is 0.  The kernel is restated as is; compare those entries against -k r.
                    zero = prm[:, 1] == 0
                    r = np.linalg.norm(x[idxs[zero, 0]] - x[idxs[zero, 1]], axis=1)
                    ref_dp[zero, 1] = -prm[zero, 0] * r
                np.testing.assert_allclose(du_dp, ref_dp, rtol=rt * 10, atol=rt * 100)
            again = impl.execute(x, prm, box, cx, cp, cu)
            np.testing.assert_array_equal(du_dx, again[0])
            np.testing.assert_array_equal(du_dp, again[1])
        # reversing the index order of every term gives identical bits (tests/test_bonded.py:109-120, test_bonded_stable.py:37-56)
        if key in ("bond", "angle"):
            rev = cls(np.ascontiguousarray(idxs[:, ::-1])).to_gpu(precision).unbound_impl
            a = impl.execute_raw(x, prm, box)
            b = rev.execute_raw(x, prm, box)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            assert a[2] == b[2]
    # wrong parameter count -> the reference's message
    with pytest.raises(RuntimeError, match=r"HarmonicBond::execute_device\(\): expected P == 2\*B, got P=3, 2\*B=80"):
        P.HarmonicBond(g["bond_idxs"]).to_gpu(precision).unbound_impl.execute(x, np.zeros(3), box)
    # empty term lists are legal and contribute nothing
    empty = P.HarmonicBond(np.zeros((0, 2), dtype=np.int32)).to_gpu(precision).unbound_impl
    du_dx, du_dp, u = empty.execute(x, np.zeros((0, 2)), box)
    assert np.all(du_dx == 0) and du_dp.shape == (0, 2) and u == 0.0


# ----------------------------------------------------------------------------------------------------------------
# Hilbert sort + neighbor list (integer / index work: exact)
# ----------------------------------------------------------------------------------------------------------------
def test_hilbert_sort_permutation_is_bit_exact(co):
    g = load("hilbert.npz")
    water = np.load(os.path.join(GOLDEN, "water.npy"))[:, :3]
    perm = co.HilbertSort(water.shape[0]).sort(water, g["box"])
    np.testing.assert_array_equal(perm, g["perm"])
    # ties: many atoms in one bin must keep their input order (stable sort)
    from oracle import hilbert as ohilbert

    rng = np.random.default_rng(3)
    x = rng.uniform(0, 0.2, (5000, 3)) + np.array([1.0, 2.0, 3.0])
    box = np.eye(3) * 6.0
    perm2 = co.HilbertSort(5000).sort(x, box)
    assert len(np.unique(ohilbert.keys(x, box))) < 200
    np.testing.assert_array_equal(perm2, ohilbert.sort_perm(x, box))
    with pytest.raises(RuntimeError, match="number of idxs to sort must be less than or equal to N"):
        co.HilbertSort(10).sort(x, box)


@pytest.mark.parametrize("size", [12, 128, 156, 298])
def test_block_bounds(co, size):
    from oracle import nblist as onblist

    rng = np.random.default_rng(2020)
    coords = rng.normal(size=(size, 3))
    box = np.eye(3) * (rng.uniform(size=3) + 1)
    for cls, real, tol in ((co.Neighborlist_f32, np.float32, 1e-6), (co.Neighborlist_f64, np.float64, 1e-7)):
        ctr, ext = cls(size).compute_block_bounds(coords, box, 32)
        rc, re = onblist.block_bounds(coords, box, real=real)
        np.testing.assert_allclose(ctr, rc, atol=tol, rtol=tol)
        np.testing.assert_allclose(ext, re, atol=tol, rtol=tol)
        with pytest.raises(RuntimeError, match="Block size must be 32."):
            cls(size).compute_block_bounds(coords, box, 64)


@pytest.mark.parametrize("num_atoms", [35, 64, 129, 1025, 1259, 2029])
def test_neighborlist_matches_brute_force(co, num_atoms):
    """tests/test_nblist.py:236-265: per 32-row block, the reported column set equals brute force at the list cutoff."""
    from oracle import hilbert as ohilbert
    from oracle import nblist as onblist

    water = np.load(os.path.join(GOLDEN, "water.npy")).astype(np.float32).astype(np.float64)[:, :3]
    rng = np.random.default_rng(1234)
    coords = water[rng.choice(num_atoms, num_atoms, replace=False)]
    box = np.eye(3) * (coords.max(0) - coords.min(0) + 0.1)
    coords = coords[ohilbert.sort_perm(coords, box)]
    ref = onblist.brute_force_ixn_list(coords, box, 1.0)
    for cls in (co.Neighborlist_f32, co.Neighborlist_f64):
        nb = cls(num_atoms)
        for _ in range(2):
            test = nb.get_nblist(coords, box, 1.0)
            assert len(test) == len(ref)
            for a, b in zip(ref, test):
                assert sorted(a) == sorted(b)
        assert nb.get_tile_ixn_count() == onblist.tile_count(ref)
        assert nb.get_num_row_idxs() == num_atoms
        nblocks = (num_atoms + 31) // 32
        assert nb.get_max_ixn_count() == nblocks * (nblocks + 1) // 2 * 32
    # row subset vs complement columns (tests/test_nblist.py:188-233)
    rows = rng.choice(num_atoms, num_atoms // 2, replace=False).astype(np.uint32)
    ref_rows = onblist.brute_force_ixn_list_rows(coords, box, 1.0, rows)
    nb = co.Neighborlist_f64(num_atoms)
    nb.set_row_idxs(rows)
    test = nb.get_nblist(coords, box, 1.0)
    assert len(test) == len(ref_rows)
    for a, b in zip(ref_rows, test):
        assert sorted(a) == sorted(b)
    nb.reset_row_idxs()
    for a, b in zip(ref, nb.get_nblist(coords, box, 1.0)):
        assert sorted(a) == sorted(b)


def test_neighborlist_validation_messages(co):
    nb = co.Neighborlist_f32(3)
    with pytest.raises(RuntimeError, match="size is greater than max size: 4 > 3"):
        nb.resize(4)
    with pytest.raises(RuntimeError, match="size is must be at least 1"):
        nb.resize(0)
    with pytest.raises(RuntimeError, match="idxs can't be empty"):
        nb.set_row_idxs(np.zeros(0, dtype=np.uint32))
    with pytest.raises(RuntimeError, match="atom indices must be unique"):
        nb.set_row_idxs(np.array([1, 1], dtype=np.uint32))
    with pytest.raises(RuntimeError, match="number of idxs must be less than N"):
        nb.set_row_idxs(np.array([0, 1, 2], dtype=np.uint32))
    with pytest.raises(RuntimeError, match="indices values must be less than N"):
        nb.set_row_idxs(np.array([7], dtype=np.uint32))


def test_all_pairs_validation_messages(co, P):
    # tests/nonbonded/test_nonbonded_all_pairs.py:10-50
    with pytest.raises(RuntimeError, match="indices can't be empty"):
        co.NonbondedAllPairs_f32(3, 2.0, 1.1, np.zeros(0, dtype=np.int32))
    with pytest.raises(RuntimeError, match=r"index values must be less than N\(3\)"):
        co.NonbondedAllPairs_f32(3, 2.0, 1.1, np.array([0, 5], dtype=np.int32))
    with pytest.raises(RuntimeError, match="index values must be greater or equal to zero"):
        co.NonbondedAllPairs_f32(3, 2.0, 1.1, np.array([-1, 1], dtype=np.int32))
    pot = co.NonbondedAllPairs_f32(1, 2.0, 1.1)
    with pytest.raises(RuntimeError, match=r"NonbondedAllPairs::execute_device\(\): expected N == N_, got N=2, N_=1"):
        pot.execute(np.zeros((2, 3)), np.zeros((2, 4)), np.eye(3) * 3)
    with pytest.raises(RuntimeError, match=r"NonbondedAllPairs::execute_device\(\): expected P == N_\*4, got P=6, N_\*4=4"):
        pot.execute(np.zeros((1, 3)), np.zeros((1, 6)), np.eye(3) * 3)
    with pytest.raises(RuntimeError, match="box must be ortholinear"):
        pot.execute(np.zeros((1, 3)), np.zeros((1, 4)), np.ones((3, 3)))
    with pytest.raises(RuntimeError, match="coords dimensions must be 2"):
        pot.execute(np.zeros((1, 1, 3)), np.zeros((1, 4)), np.eye(3))
    # K == 1 -> exactly zero (test_nonbonded_all_pairs.py:74-90)
    du_dx, du_dp, u = pot.execute(np.zeros((1, 3)), np.ones((1, 4)), np.eye(3) * 3)
    assert np.all(du_dx == 0) and np.all(du_dp == 0) and u == 0
    # set_atom_idxs == fresh object, bitwise
    rng = np.random.default_rng(2)
    N = 70
    x = rng.uniform(0, 3, (N, 3))
    p = np.stack([rng.normal(size=N), rng.uniform(0.05, 0.15, N), rng.uniform(0.2, 1, N), np.zeros(N)], 1)
    box = np.eye(3) * 3
    sub = np.sort(rng.choice(N, 30, replace=False)).astype(np.int32)
    a = co.NonbondedAllPairs_f64(N, 2.0, 1.2)
    a.execute(x, p, box)
    a.set_atom_idxs(sub)
    assert a.get_num_atom_idxs() == 30 and a.get_atom_idxs() == sub.tolist()
    b = co.NonbondedAllPairs_f64(N, 2.0, 1.2, sub)
    ra, rb = a.execute_raw(x, p, box), b.execute_raw(x, p, box)
    np.testing.assert_array_equal(ra[0], rb[0])
    np.testing.assert_array_equal(ra[1], rb[1])
    assert ra[2] == rb[2]


# ----------------------------------------------------------------------------------------------------------------
# Potential plumbing
# ----------------------------------------------------------------------------------------------------------------
def test_bound_potential_and_batches(co, P):
    from timemachine_amd import testsystems as ts

    s = ts.add_chain_ligand(ts.build_water_box(216, 2.7, seed=9), 12, lamb=0.2)
    N = s.num_atoms
    nb = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(np.float64)
    bound = nb.bind(s.nb_params).bound_impl
    assert bound.size() == N * 4 and bound.get_potential() is nb.unbound_impl
    du_dx, u = bound.execute(s.coords, s.box)
    ref = nb.unbound_impl.execute(s.coords, s.nb_params, s.box)
    np.testing.assert_array_equal(du_dx, ref[0])
    assert u == ref[2]
    with pytest.raises(RuntimeError, match="parameter size is not equal to device buffer size"):
        bound.set_params(np.zeros(3))
    p2 = s.nb_params.copy()
    p2[:, 0] *= 0.5
    bound.set_params(p2)
    assert bound.execute(s.coords, s.box)[1] == nb.unbound_impl.execute(s.coords, p2, s.box)[2]
    # execute_batch: [C, Pb, ...] outer loop coords, inner loop params
    rng = np.random.default_rng(0)
    coords = np.stack([s.coords, s.coords + rng.normal(scale=0.002, size=s.coords.shape)])
    boxes = np.stack([s.box, s.box])
    params = np.stack([s.nb_params, p2, s.nb_params * np.array([1.0, 1.0, 0.5, 1.0])])
    bx, bp, bu = nb.unbound_impl.execute_batch(coords, params, boxes, True, True, True)
    assert bx.shape == (2, 3, N, 3) and bp.shape == (2, 3, N, 4) and bu.shape == (2, 3)
    for i, j in itertools.product(range(2), range(3)):
        e = nb.unbound_impl.execute(coords[i], params[j], boxes[i])
        np.testing.assert_array_equal(bx[i, j], e[0])
        np.testing.assert_array_equal(bp[i, j], e[1])
        assert bu[i, j] == e[2]
    assert nb.unbound_impl.execute_batch(coords, params, boxes, False, False, True)[:2] == (None, None)
    ci = np.array([1, 0, 1], dtype=np.uint32)
    pi = np.array([2, 0, 1], dtype=np.uint32)
    sx, sp, su = nb.unbound_impl.execute_batch_sparse(coords, params, boxes, ci, pi, True, True, True)
    for k in range(3):
        np.testing.assert_array_equal(sx[k], bx[ci[k], pi[k]])
        np.testing.assert_array_equal(sp[k], bp[ci[k], pi[k]])
        assert su[k] == bu[ci[k], pi[k]]
    with pytest.raises(RuntimeError, match="coords_batch_idxs contains an index that is out of bounds"):
        nb.unbound_impl.execute_batch_sparse(coords, params, boxes, np.array([2], dtype=np.uint32), np.array([0], dtype=np.uint32), True, True, True)
    cb_x, cb_u = bound.execute_batch(coords, boxes, True, True)
    np.testing.assert_array_equal(cb_x[0], bx[0, 1])
    assert cb_u[1] == bu[1, 1]
    # the python front-ends (reference: potentials/potential.py) give the same energy
    assert nb(s.coords, s.nb_params, s.box) == ref[2]
    assert P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).bind(s.nb_params).to_gpu(np.float64)(s.coords, s.box) == ref[2]


# ----------------------------------------------------------------------------------------------------------------
# Integrator + Context
# ----------------------------------------------------------------------------------------------------------------
def _md_system():
    from timemachine_amd import testsystems as ts

    return ts.add_chain_ligand(ts.build_water_box(300, 3.0, seed=4), 16, lamb=0.0)


def test_context_deterministic_steps_match_oracle(co, P):
    """friction = 0 => no noise: 12 steps of ctxt.step() against the oracle's model of the device arithmetic, with forces
    taken from the GPU's own fixed-point output; and against the f64 python BAOAB with oracle forces (loose: the
    bound integrator rounds velocities to float like the reference's, wrap_kernels.cpp:700).  tests/test_md.py:142-247."""
    from oracle import integrator as oi
    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = _md_system()
    N = s.num_atoms
    rng = np.random.default_rng(2)
    v0 = rng.normal(size=(N, 3)) * 0.2
    dt, T = 1.0e-3, 300.0
    bps64 = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s)]
    intg = LangevinIntegrator(T, dt, 0.0, s.masses, 2024).impl()
    ctxt = co.Context(s.coords, v0, s.box, intg, bps64)
    ca, cb, cc, dt_r = oi.device_coefficients(T, dt, 0.0, s.masses)
    assert np.all(cc == 0)
    x, v = s.coords.copy(), v0.copy()
    xf, vf = s.coords.copy(), v0.copy()
    caf, cbf, ccf = oi.langevin_coefficients(T, dt, 0.0, s.masses)
    eval_bps = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s)]
    for step in range(12):
        # forces at the model's current x from fresh evaluations (same kernels => same fixed-point values)
        fixed = np.zeros((N, 3), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for bp in eval_bps:
                dx, _ = bp.execute(x, s.box, True, False)
                fixed += np.rint(dx * 2.0**36).astype(np.int64).view(np.uint64)
        x, v = oi.baoab_step_device_model(x, v, fixed, np.zeros((N, 3)), ca, cb, cc, dt_r)
        ctxt.step()
        np.testing.assert_allclose(ctxt.get_x_t(), x, rtol=0, atol=1e-12)
        np.testing.assert_allclose(ctxt.get_v_t(), v, rtol=0, atol=1e-7)
        if step < 4:  # independent f64 path with oracle forces
            f = np.zeros((N, 3))
            f -= rp.nonbonded(xf, s.nb_params, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)[1]
            f -= rp.harmonic_bond(xf, s.bond_params, s.box, s.bond_idxs)[1]
            f -= rp.harmonic_angle(xf, s.angle_params, s.box, s.angle_idxs)[1]
            f -= rp.periodic_torsion(xf, s.torsion_params, s.box, s.torsion_idxs)[1]
            xf, vf = oi.baoab_step(xf, vf, f, np.zeros((N, 3)), caf, cbf, ccf, dt)
            np.testing.assert_allclose(ctxt.get_x_t(), xf, rtol=0, atol=5e-7)
            np.testing.assert_allclose(ctxt.get_v_t(), vf, rtol=0, atol=2e-4)
    np.testing.assert_array_equal(ctxt.get_box(), s.box)


def test_context_multiple_steps_semantics(co, P):
    """store_x_interval semantics, setters/getters, validation errors (tests/test_md.py:24-140,895-1009)."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = _md_system()
    N = s.num_atoms
    v0 = np.zeros((N, 3))

    def make(seed=2024, friction=1.0, dt=0.5e-3):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
        return co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, seed).impl(), bps)

    c1, c2 = make(), make()
    xs, boxes = c1.multiple_steps(20, 5)
    assert xs.shape == (4, N, 3) and boxes.shape == (4, 3, 3)
    np.testing.assert_array_equal(xs[-1], c1.get_x_t())
    # same seed => same trajectory, bit for bit (counter-based noise); stepping one at a time is the same thing
    for k in range(20):
        c2.step()
        if (k + 1) % 5 == 0:
            np.testing.assert_array_equal(c2.get_x_t(), xs[(k + 1) // 5 - 1])
    np.testing.assert_array_equal(c1.get_v_t(), c2.get_v_t())
    c3 = make(seed=7)
    xs3, _ = c3.multiple_steps(20, 0)
    assert xs3.shape == (1, N, 3) and not np.array_equal(xs3[0], xs[-1])
    assert np.all(np.isfinite(xs3))
    xs_none, boxes_none = make().multiple_steps(5, 10)  # interval > n_steps: no frames, no box check
    assert xs_none.shape == (0, N, 3) and boxes_none.shape == (0, 3, 3)
    with pytest.raises(RuntimeError, match="store_x_interval must be greater than or equal to zero"):
        c1.multiple_steps(5, -1)
    # setters / getters round trip
    newx = s.coords + 0.001
    c1.set_x_t(newx)
    c1.set_v_t(v0 + 0.5)
    np.testing.assert_array_equal(c1.get_x_t(), newx)
    np.testing.assert_array_equal(c1.get_v_t(), v0 + 0.5)
    with pytest.raises(RuntimeError, match="number of new coords disagree with current coords"):
        c1.set_x_t(newx[:-1])
    with pytest.raises(RuntimeError, match="box must be 3x3"):
        c1.set_box(np.eye(4))
    assert c1.get_integrator() is not None and len(c1.get_potentials()) == 4 and c1.get_movers() == [] and c1.get_barostat() is None
    # box smaller than 2 * (cutoff + padding) is rejected when a frame is collected
    small = make()
    small.set_box(np.eye(3) * 2.5)
    with pytest.raises(RuntimeError, match="cutoff with padding is more than half of the box width, neighborlist is no longer reliable"):
        small.multiple_steps(2, 1)
    # exploded coordinates are rejected
    boom = make()
    far = s.coords.copy()
    far[0] += 1e4
    boom.set_x_t(far)
    with pytest.raises(RuntimeError, match="simulation unstable: dimensions of coordinates two orders of magnitude larger than max box dimension"):
        boom.multiple_steps(1, 1)
    with pytest.raises(RuntimeError, match="v0 N != x0 N"):
        co.Context(s.coords, v0[:-1], s.box, LangevinIntegrator(300.0, 1e-3, 1.0, s.masses, 1).impl(), [])


@pytest.mark.parametrize("precision", [np.float32, np.float64])
def test_pregathered_steps_are_bitwise_the_full_path(co, P, precision):
    """On MD steps the integrator's update kernel leaves the nonbonded gather done (positions into the sorted records,
    rebuild test, accumulator zeroed) and check+gather is not launched.  Everything that can make those inputs stale has
    to drop them: over 330 steps (rebuilds every few steps, three Hilbert re-sorts) a context stepped in one go, a
    context whose coordinates are re-set before every step (always the full path), one whose potentials are also
    evaluated from outside between steps, and one whose parameters are swapped and swapped back must agree bit for bit."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = _md_system()
    N = s.num_atoms
    rng = np.random.default_rng(11)
    v0 = rng.normal(size=(N, 3)) * 0.3

    def make():
        bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s)]
        return co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 99).impl(), bps), bps

    n_steps = 330
    ref, _ = make()
    xs_ref, _ = ref.multiple_steps(n_steps, 10)

    full, _ = make()  # set_x_t invalidates the pre-gathered inputs: every step gathers itself
    outside, bps_o = make()  # an external call into the shared potentials between steps
    swapped, bps_s = make()  # parameters replaced behind the same device pointer, then restored
    nb_idx = [i for i, bp in enumerate(ts.bound_potentials(s)) if type(bp.potential).__name__ == "Nonbonded"][0]
    nb_params = np.asarray(ts.bound_potentials(s)[nb_idx].params)
    for k in range(n_steps):
        full.set_x_t(full.get_x_t())
        full.step()
        if k % 7 == 3:
            bps_o[nb_idx].execute(s.coords + 0.01, s.box, True, True)
        outside.step()
        if k % 11 == 5:
            bps_s[nb_idx].set_params(nb_params * 0.5)
            bps_s[nb_idx].set_params(nb_params)
        swapped.step()
        if (k + 1) % 10 == 0:
            frame = xs_ref[(k + 1) // 10 - 1]
            np.testing.assert_array_equal(full.get_x_t(), frame)
            np.testing.assert_array_equal(outside.get_x_t(), frame)
            np.testing.assert_array_equal(swapped.get_x_t(), frame)
    np.testing.assert_array_equal(full.get_v_t(), ref.get_v_t())
    # a parameter change that is NOT undone must show (the stale gathered records would hide it)
    changed, bps_c = make()
    changed.multiple_steps(20, 0)
    bps_c[nb_idx].set_params(nb_params * np.array([0.0, 1.0, 1.0, 1.0]))  # charges off
    changed.multiple_steps(20, 0)
    ref2, _ = make()
    ref2.multiple_steps(40, 0)
    assert not np.array_equal(changed.get_x_t(), ref2.get_x_t())


@pytest.mark.parametrize("precision", [np.float32, np.float64])
@pytest.mark.parametrize("static_k,barostat,merge", [(None, False, True), (0, False, True), (0, True, True), (0, True, False), (None, True, True)])
def test_pregather_with_atom_subset_and_interaction_group(co, P, precision, static_k, barostat, merge):
    """The integrator's hand-over with potentials that own only part of the atoms: an all-pairs potential over a subset
    (atoms outside it have no slot) next to an interaction group (rows | columns sorted separately) -- two producers of
    deferred forces in one context, or, merged, one carrier that does not cover every atom (the update kernel's atom-order
    form).  Stepping in one go and re-setting the coordinates before every step (always the full gather path) must agree
    bit for bit across two Hilbert re-sorts: in both precisions, on the static complete list (the default at this size) and
    on the listed pipeline, with a Monte Carlo barostat every 5 steps (reference-shaped attempts: the carrier does not
    cover every atom, so its list cannot vouch for a proposal), producers merged and not."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = _md_system()
    N = s.num_atoms
    rng = np.random.default_rng(3)
    v0 = rng.normal(size=(N, 3)) * 0.1
    ligand = np.arange(N - 16, N, dtype=np.int32)
    env = np.arange(0, N - 16, dtype=np.int32)
    subset = env[: (len(env) // 9) * 6]  # two thirds of the solvent (whole waters): the rest has no nonbonded partner at all
    in_subset = np.isin(s.exclusion_idxs, subset).all(axis=1)
    excl_idxs, excl_scales = s.exclusion_idxs[in_subset], s.scale_factors[in_subset]
    groups = [list(range(3 * k, 3 * k + 3)) for k in range((N - 16) // 3)] + [list(range(N - 16, N))]

    def make():
        pots = [
            P.HarmonicBond(s.bond_idxs).bind(s.bond_params),
            P.HarmonicAngle(s.angle_idxs).bind(s.angle_params),
            P.NonbondedAllPairs(N, s.beta, s.cutoff, atom_idxs=subset).bind(s.nb_params),
            P.NonbondedInteractionGroup(N, ligand, s.beta, s.cutoff, col_atom_idxs=subset).bind(s.nb_params),
            P.NonbondedExclusions(excl_idxs, excl_scales, s.beta, s.cutoff).bind(s.nb_params),
        ]
        bps = [bp.to_gpu(precision).bound_impl for bp in pots]
        movers = [MonteCarloBarostat(N, 1.0, 300.0, groups, 5, 23).impl(bps)] if barostat else []
        return co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 2.0e-4, 5.0, s.masses, 17).impl(), bps, movers=movers), movers

    n_steps = 210
    m0 = co.debug_set_merge_producers(merge)
    k0 = co.debug_set_static_list_max_k(static_k) if static_k is not None else None
    try:
        ref, ref_movers = make()
        xs_ref, boxes_ref = ref.multiple_steps(n_steps, 15)
        assert np.all(np.isfinite(xs_ref))
        full, _ = make()
        for k in range(n_steps):
            full.set_x_t(full.get_x_t())
            full.step()
            if (k + 1) % 15 == 0:
                np.testing.assert_array_equal(full.get_x_t(), xs_ref[(k + 1) // 15 - 1])
                np.testing.assert_array_equal(full.get_box(), boxes_ref[(k + 1) // 15 - 1])
        np.testing.assert_array_equal(full.get_v_t(), ref.get_v_t())
        if barostat:
            attempts, fast = ref_movers[0].get_attempt_paths()
            assert (attempts, fast) == (n_steps // 5, 0)  # partial coverage: never the fast path
            assert ref_movers[0].get_counters()[0] > 0 and not np.array_equal(ref.get_box(), s.box)
    finally:
        co.debug_set_merge_producers(m0)
        if k0 is not None:
            co.debug_set_static_list_max_k(k0)


def test_langevin_thermostat_statistics(co, P):
    """Statistical parity for the stochastic part (cuRAND streams cannot be matched): ideal gas (no potentials) with
    friction reaches kT per degree of freedom; the noise has zero mean, unit variance and no lag-1 correlation."""
    from timemachine_amd.lib import LangevinIntegrator

    N = 20000
    rng = np.random.default_rng(0)
    masses = rng.uniform(1.0, 16.0, N)
    x0 = rng.uniform(0, 5, (N, 3))
    ctxt = co.Context(x0, np.zeros((N, 3)), np.eye(3) * 5, LangevinIntegrator(300.0, 2.5e-3, 50.0, masses, 99).impl(), [])
    ctxt.multiple_steps(200, 0)
    v = ctxt.get_v_t()
    kT = 0.008314462618 * 300.0
    ke_per_dof = (masses[:, None] * v * v).mean()
    assert abs(ke_per_dof / kT - 1.0) < 0.02
    z = v * np.sqrt(masses[:, None] / kT)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 0.03 and abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03
    from scipy import stats

    assert stats.kstest(z.reshape(-1)[:50000], "norm").pvalue > 1e-3


# ----------------------------------------------------------------------------------------------------------------
# Full size (BASELINE config 3): size-independent properties + a sampled comparison with the oracle
# ----------------------------------------------------------------------------------------------------------------
def test_dhfr_sized_box_properties(co, P):
    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts

    s = ts.dhfr_sized_water_box()
    N = s.num_atoms
    assert N == 23559
    x, p, box = s.coords, s.nb_params, s.box
    nb = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(np.float64).unbound_impl
    a = nb.execute_raw(x, p, box)
    b = nb.execute_raw(x, p, box)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[2] == b[2]
    with np.errstate(over="ignore"):
        assert np.all(a[0].sum(axis=0, dtype=np.uint64) == 0)  # Newton's third law, exactly
    nohilb = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, disable_hilbert_sort=True).to_gpu(np.float64).unbound_impl
    c = nohilb.execute_raw(x, p, box)
    np.testing.assert_array_equal(a[0], c[0])
    np.testing.assert_array_equal(a[1], c[1])
    assert a[2] == c[2]
    # oracle forces for a sample of atoms: F_i = sum_j dU_ij/dx_i over every j outside i's own (fully excluded) water
    import torch

    rng = np.random.default_rng(1)
    sample = np.sort(rng.choice(N, 256, replace=False))
    pt = torch.tensor(p)
    bt = torch.tensor(np.diagonal(box).copy())
    others = torch.arange(N)
    du_dx = a[0].view(np.int64).astype(np.float64) / 2.0**36
    ref = np.zeros((len(sample), 3))
    xt2 = torch.tensor(x)
    for k0 in range(0, len(sample), 64):
        idx = sample[k0 : k0 + 64]
        xi = torch.tensor(x[idx], requires_grad=True)
        d3 = rp.delta_r(xi[:, None, :], xt2[None, :, :], bt)
        d2 = (d3 * d3).sum(-1)
        m = torch.tensor(idx)[:, None] != others[None, :]
        ex_mask = torch.ones_like(m)
        for r, i in enumerate(idx):
            w0 = (i // 3) * 3
            ex_mask[r, w0 : w0 + 3] = False  # same water molecule: excluded (scale 1) or self
        d2 = torch.where(m & ex_mask, d2, torch.full_like(d2, 1e6))
        lj, es = rp._pair_energies(torch.sqrt(d2), pt[idx, 0][:, None] * pt[None, :, 0], pt[idx, 1][:, None] + pt[None, :, 1], pt[idx, 2][:, None] * pt[None, :, 2], s.beta, s.cutoff)
        ref[k0 : k0 + 64] = torch.autograd.grad((lj + es).sum(), xi)[0].numpy()
    assert_equal_vectors(ref, du_dx[sample], 1e-8)


@pytest.mark.parametrize("precision,cutoff", [(np.float64, 1.2), (np.float32, 1.2), (np.float64, 1.0), (np.float32, 1.0)])
def test_dhfr_shaped_box_all_terms(co, P, precision, cutoff):
    """The bench workload (testsystems.dhfr_shaped_box: 7 023 waters + a 2 490-atom solute with every bonded term kind and
    1-4 exclusions at partial scales), coordinates jittered by 0.004 nm so that every term pulls: the bonded terms against the
    oracle over the whole system, the Nonbonded force on 256 sampled atoms (half of them solute atoms, whose exclusions carry
    the partial scales) against the oracle's pair function with the reference's semantics -- all pairs minus scale x pair
    (potentials/nonbonded.py:221-399) -- and the size-independent properties of test_dhfr_sized_box_properties.
    In both precisions and at both cutoffs the bench line quotes: rc 1.2 (BASELINE config 3) and rc 1.0 / f32, the reference's own
    dhfr-apo configuration (testsystems/dhfr.py:21, tests/test_benchmark.py:219); f32 tolerance: 1e-4 of the force norm, the
    reference's (tests/common.py:250-334).  The f32 kernels have their own workgroup shape and pool split at this size."""
    import torch

    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts

    s = ts.dhfr_shaped_box(cutoff=cutoff)
    N = s.num_atoms
    assert N == 23559 and len(s.torsion_idxs) == 8610 and np.sum(s.scale_factors[:, 0] != 1.0) == 7020 and s.cutoff == cutoff
    rng = np.random.default_rng(11)
    x = s.coords + rng.normal(0.0, 0.004, s.coords.shape)
    if precision == np.float32:
        x = x.astype(np.float32).astype(np.float64)  # both sides see the coordinates the f32 kernels see
    p, box = s.nb_params, s.box
    f64 = precision == np.float64
    if cutoff == 1.2:
        for cls, idxs, prm, ref_fn in (
            (P.HarmonicBond, s.bond_idxs, s.bond_params, rp.harmonic_bond),
            (P.HarmonicAngle, s.angle_idxs, s.angle_params, rp.harmonic_angle),
            (P.PeriodicTorsion, s.torsion_idxs, s.torsion_params, rp.periodic_torsion),
        ):
            du_dx, du_dp, u = cls(idxs).to_gpu(precision).unbound_impl.execute(x, prm, box)
            ref_u, ref_dx, ref_dp = ref_fn(x, prm, box, idxs)
            brt = 1e-7 if f64 else 1e-4
            assert abs(u - ref_u) <= (0.1 * brt) * max(1.0, abs(ref_u)) if f64 else abs(u - ref_u) <= brt * max(1.0, abs(ref_u)), cls.__name__
            if f64:
                assert_equal_vectors(ref_dx, du_dx, brt)
            else:
                # f32: the absolute error of a stiff bond is k * eps_f32 * r ~ 4.6e5 * 6e-8 * 0.1 = 3e-3 kJ/mol/nm whatever the
                # arithmetic, and among 23.5k atoms some have bonded forces that cancel to < 1 kJ/mol/nm (measured: 7.0e-3 absolute
                # on an atom whose bond forces sum to 0.34): the error is taken relative to max(|F|, 100) -- the terms pull with
                # ~1e3 on these strained coordinates
                norms = np.maximum(np.linalg.norm(ref_dx, axis=1, keepdims=True), 100.0)
                assert (np.abs(ref_dx - du_dx) / norms).max() <= brt, cls.__name__
            assert np.abs(du_dp - ref_dp).max() <= brt * max(1.0, np.abs(ref_dp).max()), cls.__name__
            assert np.linalg.norm(ref_dx[s.num_water_atoms :], axis=1).max() > 100.0  # the solute's terms do pull
    nb = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(precision).unbound_impl
    a = nb.execute_raw(x, p, box)
    b = nb.execute_raw(x, p, box)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[2] == b[2]
    with np.errstate(over="ignore"):
        assert np.all(a[0].sum(axis=0, dtype=np.uint64) == 0)  # Newton's third law, exactly
    c = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, disable_hilbert_sort=True).to_gpu(precision).unbound_impl.execute_raw(x, p, box)
    np.testing.assert_array_equal(a[0], c[0])
    assert a[2] == c[2]
    # forces-only (the MD form; f64: table-driven electrostatics) gives the same bits as the full call -- on the wave-per-item
    # kernel (the product) and on the row-block kernel (the second implementation of the same contract)
    for rowblock_min_k in ((2**31 - 1, 0) if co.debug_rowblock_available() else (2**31 - 1,)):  # (the variant library carries the second kernel)
        before = co.debug_set_rowblock_min_k(rowblock_min_k)
        try:
            np.testing.assert_array_equal(nb.execute_raw(x, p, box, True, False, False)[0], a[0])
        finally:
            co.debug_set_rowblock_min_k(before)

    sample = np.sort(np.concatenate([rng.choice(s.num_water_atoms, 128, replace=False), s.num_water_atoms + rng.choice(N - s.num_water_atoms, 128, replace=False)]))
    row_of = {int(i): r for r, i in enumerate(sample)}
    keep_q, keep_lj = np.ones((len(sample), N)), np.ones((len(sample), N))  # fraction of the pair that REMAINS
    for (i, j), (sq, slj) in zip(s.exclusion_idxs, s.scale_factors):
        for a_, b_ in ((int(i), int(j)), (int(j), int(i))):
            if a_ in row_of:
                keep_q[row_of[a_], b_] = 1.0 - sq
                keep_lj[row_of[a_], b_] = 1.0 - slj
    pt = torch.tensor(p)
    bt = torch.tensor(np.diagonal(box).copy())
    xt2 = torch.tensor(x)
    others = torch.arange(N)
    du_dx = a[0].view(np.int64).astype(np.float64) / 2.0**36
    ref = np.zeros((len(sample), 3))
    for k0 in range(0, len(sample), 64):
        idx = sample[k0 : k0 + 64]
        xi = torch.tensor(x[idx], requires_grad=True)
        d3 = rp.delta_r(xi[:, None, :], xt2[None, :, :], bt)
        d2 = (d3 * d3).sum(-1)
        m = torch.tensor(idx)[:, None] != others[None, :]
        d2 = torch.where(m, d2, torch.full_like(d2, 1e6))
        lj, es = rp._pair_energies(torch.sqrt(d2), pt[idx, 0][:, None] * pt[None, :, 0], pt[idx, 1][:, None] + pt[None, :, 1], pt[idx, 2][:, None] * pt[None, :, 2], s.beta, s.cutoff)
        total = (lj * torch.tensor(keep_lj[k0 : k0 + 64])).sum() + (es * torch.tensor(keep_q[k0 : k0 + 64])).sum()
        ref[k0 : k0 + 64] = torch.autograd.grad(total, xi)[0].numpy()
    assert_equal_vectors(ref, du_dx[sample], 1e-8 if f64 else 1e-4)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_group_stepping_is_bitwise_separate_stepping(co, P, precision):
    """custom_ops.multiple_steps_group steps several contexts interleaved on streams of their own (windows / HREX replicas that
    share a GPU: one context's list and update kernels run underneath another's force kernel).  The contexts share no state, so
    every trajectory must be, bit for bit, the one the same context takes through multiple_steps alone -- across a Hilbert re-sort
    (call 100), a barostat in one of the contexts, and mixed with ordinary calls before and after.  (This system is small enough
    for the STATIC complete list: no rebuild runs here -- test_group_stepping_on_the_listed_pipeline_... below is the one that
    groups contexts whose lists rebuild, at the sizes bench.py groups.)"""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    x0 = s.coords.astype(np.float32).astype(np.float64)

    def contexts():
        out = []
        for k in range(3):
            bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision)]
            movers = [MonteCarloBarostat(N, 1.0, 300.0, ts.molecule_groups(s), 10, 5).impl(bps)] if k == 1 else []
            out.append(co.Context(x0, np.zeros_like(x0), s.box, LangevinIntegrator(300.0, 1.5e-3, 2.0, s.masses, 40 + k).impl(), bps, movers=movers))
        return out

    alone, grouped = contexts(), contexts()
    for c in alone:
        c.multiple_steps(30, 0)
        c.multiple_steps(120, 0)
        c.multiple_steps(7, 0)
    for c in grouped:
        c.multiple_steps(30, 0)
    co.multiple_steps_group(grouped, 120)
    for c in grouped:
        assert c.last_multiple_steps_ms() > 0
    co.multiple_steps_group(grouped[:1], 7)  # a group of one
    co.multiple_steps_group(grouped[1:], 7)
    for a, g in zip(alone, grouped):
        np.testing.assert_array_equal(a.get_x_t(), g.get_x_t())
        np.testing.assert_array_equal(a.get_v_t(), g.get_v_t())
        np.testing.assert_array_equal(a.get_box(), g.get_box())
    assert not np.array_equal(alone[0].get_x_t(), alone[2].get_x_t())  # different seeds: different trajectories
    co.multiple_steps_group([], 5)  # nothing to do
    co.multiple_steps_group(grouped, 0)
    with pytest.raises(RuntimeError, match="distinct"):
        co.multiple_steps_group([grouped[0], grouped[0]], 1)


@pytest.mark.parametrize("threads", ["1", "2"])
def test_calls_longer_than_the_run_ahead_bound_are_bitwise_short_calls(co, P, threads, monkeypatch):
    """The stepping loops keep the host a bounded number of steps ahead of the device (csrc/integrator.hip, RunAhead: at most 64
    by the Langevin integrator's progress word; an event every 32 steps for an integrator without one -- VelocityVerlet).  A call of
    700 steps sleeps on that bound many times and must leave the state that short calls leave, bit for bit: alone (with frames:
    their copies wait as well), grouped, and under the velocity Verlet integrator."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, VelocityVerletIntegrator

    monkeypatch.setenv("TM_AMD_GROUP_THREADS", threads)
    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    x0 = s.coords.astype(np.float32).astype(np.float64)

    def contexts(n):
        out = []
        for k in range(n):
            bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s, np.float32)]
            movers = [MonteCarloBarostat(N, 1.0, 300.0, ts.molecule_groups(s), 25, 9 + k).impl(bps)]
            out.append(co.Context(x0, np.zeros_like(x0), s.box, LangevinIntegrator(300.0, 1.5e-3, 2.0, s.masses, 70 + k).impl(), bps, movers=movers))
        return out

    def verlet():
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s, np.float32)]
        return co.Context(x0, np.zeros_like(x0), s.box, VelocityVerletIntegrator(0.5e-3, s.masses).impl(), bps)

    vv_short, vv_long = verlet(), verlet()
    vv_short.initialize()  # (a call per step never meets the bound; initialize / finalize as multiple_steps places them)
    for _ in range(200):
        vv_short.step()
    vv_short.finalize()
    vv_long.multiple_steps(200, 0)
    np.testing.assert_array_equal(vv_short.get_x_t(), vv_long.get_x_t())
    np.testing.assert_array_equal(vv_short.get_v_t(), vv_long.get_v_t())

    short, long_, grouped = contexts(3), contexts(3), contexts(3)
    for c in short:
        for _ in range(7):
            c.multiple_steps(100, 0)
    frames = [c.multiple_steps(700, 350) for c in long_]
    co.multiple_steps_group(grouped, 700)
    for a, b, g, (xs, boxes) in zip(short, long_, grouped, frames):
        assert xs.shape[0] == 2
        np.testing.assert_array_equal(xs[-1], b.get_x_t())
        for other in (b, g):
            np.testing.assert_array_equal(a.get_x_t(), other.get_x_t())
            np.testing.assert_array_equal(a.get_v_t(), other.get_v_t())
            np.testing.assert_array_equal(a.get_box(), other.get_box())


@pytest.mark.parametrize("threads", ["1", "2"])
@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which", ["listed_small", "dhfr_shaped"])
def test_group_stepping_on_the_listed_pipeline_is_bitwise_separate_stepping(co, P, which, precision, threads):
    """The grouped call where bench.py runs it (`replicas_per_gpu`, `--mode hrex`): the LISTED neighbor-list pipeline -- k_find_ixns
    rebuilds, the coordinate snapshot hand-over, the cost-bucket counters and the sorted pre-gather all running concurrently on the
    group's streams --, a barostat in EVERY context (the production shape, fe/rbfe.py:113-121), one or two enqueueing host threads.
    `listed_small`: the 2 243-atom solvated ligand with the static complete list switched off (four contexts, 130 steps: a rebuild
    every few steps and the call-100 Hilbert re-sort fall inside the grouped call); `dhfr_shaped`: three contexts of the 23 559-atom
    bench workload, 120 steps.  Against the same contexts stepped alone: identical bits."""
    import os

    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    if which == "listed_small":
        s, n_ctx, n_steps, dt, friction = ts.small_solvated_ligand(lamb=0.3), 4, 130, 1.5e-3, 2.0
    else:
        s, n_ctx, n_steps, dt, friction = ts.dhfr_shaped_box(), 3, 120, 1.0e-3, 10.0
    N = s.num_atoms
    x0 = s.coords.astype(np.float32).astype(np.float64)
    groups = ts.molecule_groups(s)

    def contexts():
        out = []
        for k in range(n_ctx):
            bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision, nblist_padding=0.18 if which == "dhfr_shaped" else 0.1)]
            movers = [MonteCarloBarostat(N, 1.0, 300.0, groups, 10 + k, 5 + k).impl(bps)]
            out.append((co.Context(x0, np.zeros_like(x0), s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 70 + k).impl(), bps, movers=movers), bps, movers))
        return out

    static_before = co.debug_set_static_list_max_k(0)  # every system of this test runs the listed pipeline
    env_before = os.environ.get("TM_AMD_GROUP_THREADS")
    os.environ["TM_AMD_GROUP_THREADS"] = threads
    try:
        alone, grouped = contexts(), contexts()
        for c, _, _ in alone:
            c.multiple_steps(n_steps, 0)
            c.multiple_steps(9, 0)
        co.multiple_steps_group([c for c, _, _ in grouped], n_steps)
        co.multiple_steps_group([c for c, _, _ in grouped], 9)
        builds = []
        for (a, bps_a, mv_a), (g, bps_g, mv_g) in zip(alone, grouped):
            np.testing.assert_array_equal(a.get_x_t(), g.get_x_t())
            np.testing.assert_array_equal(a.get_v_t(), g.get_v_t())
            np.testing.assert_array_equal(a.get_box(), g.get_box())
            assert mv_a[0].get_counters() == mv_g[0].get_counters()
            assert mv_a[0].get_volume_scale_factor() == mv_g[0].get_volume_scale_factor() != 0.0  # the barostats did attempt moves inside the grouped call
            builds.append(_nonbonded_all_pairs_of(bps_g).get_build_count())
        assert min(builds) >= 4  # ... and the lists were rebuilt inside it (not a static list)
        assert not np.array_equal(alone[0][0].get_x_t(), alone[1][0].get_x_t())
    finally:
        co.debug_set_static_list_max_k(static_before)
        if env_before is None:
            del os.environ["TM_AMD_GROUP_THREADS"]
        else:
            os.environ["TM_AMD_GROUP_THREADS"] = env_before


def _nonbonded_all_pairs_of(bound_impls):
    """the NonbondedAllPairs implementation inside a list of bound potentials (directly, or as a child of a Fanout / Summed potential)"""
    def walk(p):
        name = type(p).__name__
        if name.startswith("NonbondedAllPairs"):
            return p
        if hasattr(p, "get_potentials"):
            for c in p.get_potentials():
                r = walk(c)
                if r is not None:
                    return r
        return None

    for bp in bound_impls:
        r = walk(bp.get_potential())
        if r is not None:
            return r
    raise AssertionError("no NonbondedAllPairs among the bound potentials")


def test_md_on_the_row_block_kernel_is_bitwise_md_on_the_item_kernel(co, P):
    """(variant library only: tests/test_gpu_second_binding.py)  MD with every forces-only launch on the row-block kernel against the
    same run on the wave-per-item kernel: the two implementations of the reference's tile kernel hand the integrator identical
    forces step after step -- 2 243 atoms on the listed pipeline over 120 steps, and 23 559 atoms over 30."""
    if not co.debug_rowblock_available():
        pytest.skip("the row-block kernel is in the variant library only")
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    static_before = co.debug_set_static_list_max_k(0)
    try:
        for s, n_steps, precision in ((ts.small_solvated_ligand(lamb=0.2), 120, np.float32), (ts.small_solvated_ligand(lamb=0.2), 120, np.float64), (ts.dhfr_shaped_box(), 30, np.float64)):
            x0 = s.coords.astype(np.float32).astype(np.float64)
            out = []
            for min_k in (2**31 - 1, 0):
                before = co.debug_set_rowblock_min_k(min_k)
                try:
                    bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision)]
                    c = co.Context(x0, np.zeros_like(x0), s.box, LangevinIntegrator(300.0, 1.0e-3, 5.0, s.masses, 3).impl(), bps)
                    c.multiple_steps(n_steps, 0)
                    out.append((c.get_x_t(), c.get_v_t()))
                finally:
                    co.debug_set_rowblock_min_k(before)
            np.testing.assert_array_equal(out[0][0], out[1][0])
            np.testing.assert_array_equal(out[0][1], out[1][1])
    finally:
        co.debug_set_static_list_max_k(static_before)


def test_step_replicas_groups_of_two_equal_sequential_stepping(co, P):
    """hrex.step_replicas (how bench.py --mode hrex and an HREX driver step the replicas that share a GPU) with five contexts in
    groups of two (2 + 2 + 1) against the same contexts stepped one call after the other: identical bits."""
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = _md_system()
    v0 = np.zeros_like(s.coords)

    def contexts():
        return [co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 900 + k).impl(),
                           [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]) for k in range(5)]

    a, b = contexts(), contexts()
    for c in a:
        c.multiple_steps(60, 0)
    hrex.step_replicas(b, 60, group=2)
    for ca, cb in zip(a, b):
        np.testing.assert_array_equal(ca.get_x_t(), cb.get_x_t())
        np.testing.assert_array_equal(ca.get_v_t(), cb.get_v_t())


def test_group_stepping_refuses_contexts_that_share_device_state(co, P):
    """One unbound potential bound twice keeps ONE neighbor list and ONE set of accumulators: its two BoundPotentials may be used one
    call after the other (the reference's pattern) but not on two streams at once -- multiple_steps_group says so instead of
    computing garbage; the same contexts stepped one by one are fine."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = _md_system()
    v0 = np.zeros_like(s.coords)
    nb = P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(np.float32)
    b1, b2 = nb.bind(s.nb_params).bound_impl, nb.bind(s.nb_params * np.array([0.9, 1.0, 1.0, 1.0])).bound_impl
    c1 = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 1).impl(), [b1])
    c2 = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 2).impl(), [b2])
    with pytest.raises(RuntimeError, match="share a potential"):
        co.multiple_steps_group([c1, c2], 5)
    c1.multiple_steps(5, 0)
    c2.multiple_steps(5, 0)
    assert np.all(np.isfinite(c1.get_x_t())) and np.all(np.isfinite(c2.get_x_t()))
    shared_intg = LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 3).impl()
    fresh = [[bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)] for _ in range(2)]
    c3, c4 = (co.Context(s.coords, v0, s.box, shared_intg, fresh[k]) for k in range(2))
    with pytest.raises(RuntimeError, match="share a potential, integrator or mover"):
        co.multiple_steps_group([c3, c4], 5)
    # a barostat evaluates ITS bound potentials: one built on another context's list would run in that context's neighbor list
    from timemachine_amd.lib import MonteCarloBarostat

    own = [[bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)] for _ in range(2)]
    foreign_barostat = MonteCarloBarostat(s.num_atoms, 1.0, 300.0, ts.molecule_groups(s), 5, 3).impl(own[1])
    c5 = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 4).impl(), own[0], movers=[foreign_barostat])
    c6 = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 5).impl(), own[1])
    with pytest.raises(RuntimeError, match="share a potential, integrator or mover"):
        co.multiple_steps_group([c5, c6], 5)
    # hrex.step_replicas falls back to one call after the other for such a group instead of failing
    from timemachine_amd import hrex

    hrex.step_replicas([c5, c6], 5, group=2)
    hrex.step_replicas([], 5)
    assert np.all(np.isfinite(c5.get_x_t())) and np.all(np.isfinite(c6.get_x_t()))
