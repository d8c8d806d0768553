"""GPU tests (``-m gpu``) of the Potential / BoundPotential / SummedPotential / FanoutSummedPotential surface, case by case
after the reference's tests/test_potentials.py (:36-205 ownership, set_params, validation messages, nesting; :209-391 batch
shapes / None / equality with the one-by-one calls; :393-466 summed, fanout, bound == unbound; :467-652 sparse batches).
Every call goes Python -> ctypes -> C ABI -> HIP kernels; the oracle is only the checker for the summed potential's values."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


@pytest.fixture(scope="module")
def P():
    from timemachine_amd import potentials

    return potentials


@pytest.fixture
def harmonic_bond(P):
    bond_idxs = np.array([[0, 1], [0, 2]], dtype=np.int32)
    params = np.ones(shape=(2, 2), dtype=np.float32)
    return P.HarmonicBond(bond_idxs).bind(params)


def execute_bound_impl(bp):
    coords = np.zeros(shape=(3, 3), dtype=np.float32)
    box = np.diag(np.ones(3))
    return bp.execute(coords, box)


def test_bound_potential_keeps_referenced_potential_alive(co, harmonic_bond):
    bp = harmonic_bond.to_gpu(np.float32).bound_impl  # the wrapper objects die here; the native potential must not
    import gc

    gc.collect()
    execute_bound_impl(bp)


def test_bound_potential_get_potential(co, harmonic_bond):
    unbound_impl = harmonic_bond.potential.to_gpu(np.float32).unbound_impl
    bound_impl = co.BoundPotential(unbound_impl, harmonic_bond.params)
    assert unbound_impl is bound_impl.get_potential()


def test_bound_potential_empty_params(co, P):
    bond_idxs = np.empty((0, 2), dtype=np.int32)
    params = np.empty((0, 2))
    u_test = P.HarmonicBond(bond_idxs).bind(params).to_gpu(np.float32)
    x = np.empty((0, 3))
    box = np.eye(3)
    assert u_test.bound_impl.execute(x, box, False, True)[1] == 0.0
    u_test.bound_impl.set_params(np.empty((0, 2)))
    assert u_test.bound_impl.execute(x, box, False, True)[1] == 0.0


def test_bound_potential_set_params(co, harmonic_bond):
    x, box = np.ones((3, 3)), np.eye(3)
    x[1] += (0.3, 0.1, 0.0)
    x[2] += (0.0, 0.2, 0.4)
    new_params = np.random.default_rng(2023).random(size=(2, 2), dtype=np.float32)
    u_ref = harmonic_bond.potential.to_gpu(np.float32).bind(new_params).bound_impl
    u_test = harmonic_bond.to_gpu(np.float32).bound_impl
    assert u_test.execute(x, box)[1] != u_ref.execute(x, box)[1]
    u_test.set_params(new_params)
    assert u_test.execute(x, box)[1] == u_ref.execute(x, box)[1]
    np.testing.assert_array_equal(u_test.execute(x, box)[0], u_ref.execute(x, box)[0])
    with pytest.raises(RuntimeError, match="2 != 4"):
        u_test.set_params(np.ones(shape=(1, 2), dtype=np.float32))


def verify_potential_validation(potential):
    with pytest.raises(RuntimeError, match="coords dimensions must be 2"):
        potential(np.zeros(1), np.ones((3, 3)))
    with pytest.raises(RuntimeError, match="coords must have a shape that is 3 dimensional"):
        potential(np.zeros((1, 4)), np.ones((3, 3)))
    with pytest.raises(RuntimeError, match="box must be 3x3"):
        potential(np.zeros((1, 3)), np.ones(3))
    with pytest.raises(RuntimeError, match="box must be 3x3"):
        potential(np.zeros((1, 3)), np.ones((2, 2)))
    with pytest.raises(RuntimeError, match="box must be ortholinear"):
        potential(np.zeros((1, 3)), np.ones((3, 3)))
    with pytest.raises(RuntimeError, match="box must have positive values along diagonal"):
        potential(np.zeros((1, 3)), np.eye(3) * 0.0)


def test_bound_and_unbound_potential_execute_validation(co, harmonic_bond):
    bound_impl = harmonic_bond.to_gpu(np.float32).bound_impl
    verify_potential_validation(bound_impl.execute)
    execute_bound_impl(bound_impl)
    unbound_impl = harmonic_bond.potential.to_gpu(np.float32).unbound_impl
    verify_potential_validation(lambda coords, box: unbound_impl.execute(coords, harmonic_bond.params, box))
    unbound_impl.execute(np.zeros((3, 3)), harmonic_bond.params, np.eye(3))


def test_summed_potential_construction_and_ownership(co, P, harmonic_bond):
    with pytest.raises(ValueError) as excinfo:
        P.SummedPotential([harmonic_bond], [])
    assert str(excinfo.value) == "number of potentials != number of parameter arrays"
    sp = P.SummedPotential([harmonic_bond.potential], [harmonic_bond.params]).bind(harmonic_bond.params.reshape(-1))
    execute_bound_impl(sp.to_gpu(np.float32).bound_impl)
    impls = [harmonic_bond.potential.to_gpu(np.float32).unbound_impl for _ in range(2)]
    summed_impl = co.SummedPotential(impls, [harmonic_bond.params.size] * 2)
    assert set(id(p) for p in summed_impl.get_potentials()) == set(id(p) for p in impls)


def test_summed_potential_invalid_parameters_size(co, P, harmonic_bond):
    sp = P.SummedPotential([harmonic_bond.potential], [harmonic_bond.params])
    n = harmonic_bond.params.size
    with pytest.raises(RuntimeError) as e:
        execute_bound_impl(sp.bind(np.empty(0)).to_gpu(np.float32).bound_impl)
    assert f"SummedPotential::execute_device(): expected {n} parameters, got 0" in str(e)
    with pytest.raises(RuntimeError) as e:
        execute_bound_impl(sp.bind(np.ones(n + 1)).to_gpu(np.float32).bound_impl)
    assert f"SummedPotential::execute_device(): expected {n} parameters, got {n + 1}" in str(e)


def test_summed_potential_nested(co, P, harmonic_bond):
    nested_sp = P.SummedPotential([harmonic_bond.potential], [harmonic_bond.params])
    nested_sp_params = harmonic_bond.params.flatten()
    sp = P.SummedPotential([nested_sp, harmonic_bond.potential], [nested_sp_params, harmonic_bond.params])
    sp_params = np.concatenate([nested_sp_params, harmonic_bond.params.flatten()])
    x = np.array([(0.0, 0.0, 0.0), (1.3, 0.1, 0.0), (0.0, 0.7, 0.2)])
    box = np.eye(3) * 5.0
    _, u1 = sp.bind(sp_params).to_gpu(np.float64).bound_impl.execute(x, box)
    _, u0 = harmonic_bond.to_gpu(np.float64).bound_impl.execute(x, box)
    np.testing.assert_allclose(u1, 2 * u0, rtol=1e-12)
    sp_prime = P.SummedPotential([sp, nested_sp, harmonic_bond.potential], [sp_params, nested_sp_params, harmonic_bond.params])
    sp_prime_params = np.concatenate([sp_params, nested_sp_params, harmonic_bond.params.flatten()])
    _, u2 = sp_prime.bind(sp_prime_params).to_gpu(np.float64).bound_impl.execute(x, box)
    np.testing.assert_allclose(u2, 4 * u0, rtol=1e-12)


def reference_execute_over_batch(unbound, coords, boxes, params):
    du_dx = np.empty((coords.shape[0], params.shape[0], coords.shape[1], 3))
    du_dp = np.empty((coords.shape[0], params.shape[0], *params.shape[1:]))
    u = np.empty((coords.shape[0], params.shape[0]))
    for i in range(coords.shape[0]):
        for j in range(params.shape[0]):
            du_dx[i][j], du_dp[i][j], u[i][j] = unbound.execute(coords[i], params[j], boxes[i])
    return du_dx, du_dp, u


@pytest.mark.parametrize("precision", [np.float32, np.float64])
def test_unbound_and_bound_impl_execute_batch(co, harmonic_bond, precision):
    np.random.seed(2022)
    N = 5
    coords = np.random.random((N, 3))
    perturbed_coords = coords + np.random.random(coords.shape)
    num_coord_batches, num_param_batches = 5, 3
    box = np.diag(np.ones(3))
    coords_batch = np.stack([coords, perturbed_coords] * num_coord_batches)
    boxes_batch = np.stack([box] * 2 * num_coord_batches)
    params = harmonic_bond.params
    params_batch = np.stack([params, np.random.random(params.shape)] * num_param_batches)
    unbound_gpu = harmonic_bond.potential.to_gpu(precision)
    unbound_impl = unbound_gpu.unbound_impl
    ref_du_dx, ref_du_dp, ref_u = reference_execute_over_batch(unbound_impl, coords_batch, boxes_batch, params_batch)

    with pytest.raises(RuntimeError) as e:
        unbound_impl.execute_batch(coords_batch, params_batch, boxes_batch[:num_coord_batches], True, True, True)
    assert str(e.value) == "number of batches of coords and boxes don't match"
    with pytest.raises(RuntimeError) as e:
        unbound_impl.execute_batch(coords, params_batch, box, True, True, True)
    assert str(e.value) == "coords and boxes must have 3 dimensions"
    with pytest.raises(RuntimeError) as e:
        unbound_impl.execute_batch(coords_batch, np.ones(3), boxes_batch, True, True, True)
    assert str(e.value) == "parameters must have at least 2 dimensions"

    shape_prefix = (len(coords_batch), len(params_batch))
    for compute_du_dx, compute_du_dp, compute_u in itertools.product([False, True], repeat=3):
        b_du_dx, b_du_dp, b_u = unbound_impl.execute_batch(coords_batch, params_batch, boxes_batch, compute_du_dx, compute_du_dp, compute_u)
        if compute_du_dx:
            assert b_du_dx.shape == (*shape_prefix, N, 3)
            np.testing.assert_array_equal(b_du_dx, ref_du_dx)
        else:
            assert b_du_dx is None
        if compute_du_dp:
            assert b_du_dp.shape == (*shape_prefix, *params.shape)
            np.testing.assert_array_equal(b_du_dp, ref_du_dp)
        else:
            assert b_du_dp is None
        if compute_u:
            assert b_u.shape == shape_prefix
            np.testing.assert_array_equal(b_u, ref_u)
        else:
            assert b_u is None

    # the bound form (tests/test_potentials.py:301-360)
    bound_impl = unbound_gpu.bind(params).bound_impl
    r_du_dx, _, r_u = reference_execute_over_batch(unbound_impl, coords_batch, boxes_batch, np.array([params]))
    r_du_dx, r_u = r_du_dx.squeeze(), r_u.squeeze()
    with pytest.raises(RuntimeError) as e:
        bound_impl.execute_batch(coords_batch, boxes_batch[: num_coord_batches - 1], True, True)
    assert str(e.value) == "number of batches of coords and boxes don't match"
    with pytest.raises(RuntimeError) as e:
        bound_impl.execute_batch(coords, box, True, True)
    assert str(e.value) == "coords and boxes must have 3 dimensions"
    for compute_du_dx, compute_u in itertools.product([False, True], repeat=2):
        b_du_dx, b_u = bound_impl.execute_batch(coords_batch, boxes_batch, compute_du_dx, compute_u)
        if compute_du_dx:
            assert b_du_dx.shape == (len(coords_batch), N, 3)
            np.testing.assert_array_equal(b_du_dx, r_du_dx)
        else:
            assert b_du_dx is None
        if compute_u:
            assert b_u.shape == (len(coords_batch),)
            np.testing.assert_array_equal(b_u, r_u)
        else:
            assert b_u is None


@pytest.fixture
def harmonic_bond_test_system(P):
    np.random.seed(2022)
    num_atoms, num_bonds = 10, 10
    coords = np.random.uniform(0, 1, size=(num_atoms, 3)).astype(np.float32)

    def random_bond_idxs():
        return np.array([np.random.choice(num_atoms, size=(2,), replace=False) for _ in range(num_bonds)], dtype=np.int32)

    hb1, hb2 = P.HarmonicBond(random_bond_idxs()), P.HarmonicBond(random_bond_idxs())
    return hb1, hb2, np.random.uniform(0, 1, size=(num_bonds, 2)), np.random.uniform(0, 1, size=(num_bonds, 2)), coords


@pytest.mark.parametrize("num_potentials", [1, 2, 5])
@pytest.mark.parametrize("parallel", [False, True])
def test_summed_potential_matches_the_oracle(co, P, parallel, num_potentials, harmonic_bond_test_system):
    from oracle import ref_potentials as rp

    hb, _, params, _, coords = harmonic_bond_test_system
    box = 3.0 * np.eye(3)
    params_list = [params] * num_potentials
    potential = P.SummedPotential([hb] * num_potentials, params_list, parallel)
    flat_params = np.concatenate([p.reshape(-1) for p in params_list])
    u1, du_dx1, du_dp1 = rp.harmonic_bond(coords.astype(np.float64), params, box, hb.idxs)
    for rtol, precision in [(1e-5, np.float32), (1e-10, np.float64)]:
        du_dx, du_dp, u = potential.to_gpu(precision).unbound_impl.execute(coords, flat_params, box)
        np.testing.assert_allclose(u, num_potentials * u1, rtol=rtol)
        np.testing.assert_allclose(du_dx, num_potentials * du_dx1, rtol=rtol, atol=rtol * np.abs(du_dx1).max())
        np.testing.assert_allclose(np.asarray(du_dp).reshape(num_potentials, -1), np.tile(du_dp1.reshape(1, -1), (num_potentials, 1)), rtol=rtol, atol=rtol * np.abs(du_dp1).max())


@pytest.mark.parametrize("parallel", [False, True])
def test_fanout_summed_potential_consistency(co, P, parallel, harmonic_bond_test_system):
    hb1, hb2, params, _, coords = harmonic_bond_test_system
    summed = P.SummedPotential([hb1, hb2], [params, params])
    fanout = P.FanoutSummedPotential([hb1, hb2], parallel)
    box = 3.0 * np.eye(3)
    du_dx_ref, du_dps_ref, u_ref = summed.to_gpu(np.float32).unbound_impl.execute(coords, np.concatenate([params.reshape(-1)] * 2), box)
    du_dx_test, du_dp_test, u_test = fanout.to_gpu(np.float32).unbound_impl.execute(coords, params, box)
    np.testing.assert_array_equal(du_dx_ref, du_dx_test)
    np.testing.assert_allclose(np.sum(np.asarray(du_dps_ref).reshape(2, -1), axis=0), np.asarray(du_dp_test).reshape(-1), rtol=1e-8, atol=1e-8)
    assert u_ref == u_test


@pytest.mark.parametrize("precision", [np.float32, np.float64])
def test_bound_and_unbound_execute_match(co, harmonic_bond_test_system, precision):
    hb1, _, params, _, coords = harmonic_bond_test_system
    gpu_bond = hb1.to_gpu(precision)
    box = 3.0 * np.eye(3)
    u_du_dx, _, u_u = gpu_bond.unbound_impl.execute(coords, params, box)
    b_du_dx, b_u = gpu_bond.bind(params).bound_impl.execute(coords, box)
    np.testing.assert_array_equal(u_du_dx, b_du_dx)
    assert u_u == b_u


def test_execute_batch_sparse_validation_and_values(co, harmonic_bond):
    np.random.seed(2022)
    N = 5
    coords = np.random.random((N, 3))
    box = np.diag(np.ones(3))
    coords_batch = np.stack([coords, coords + np.random.random(coords.shape)] * 3)
    boxes_batch = np.stack([box] * 6)
    params = harmonic_bond.params
    params_batch = np.stack([params, np.random.random(params.shape)] * 2)
    impl = harmonic_bond.potential.to_gpu(np.float64).unbound_impl
    ci = np.array([0, 1, 5, 2], dtype=np.uint32)
    pi = np.array([3, 0, 1, 2], dtype=np.uint32)

    def call(c=coords_batch, p=params_batch, b=boxes_batch, i=ci, j=pi):
        return impl.execute_batch_sparse(c, p, b, i, j, True, True, True)

    cases = [
        (dict(b=boxes_batch[:3]), "number of coord arrays and boxes don't match"),
        (dict(c=coords, b=box), "coords and boxes must have 3 dimensions"),
        (dict(p=np.ones(3)), "parameters must have at least 2 dimensions"),
        (dict(i=ci.reshape(2, 2)), "coords_batch_idxs and params_batch_idxs must be one-dimensional arrays"),
        (dict(j=pi.reshape(2, 2)), "coords_batch_idxs and params_batch_idxs must be one-dimensional arrays"),
        (dict(i=ci[:3]), "coords_batch_idxs and params_batch_idxs must have the same length"),
        (dict(i=np.array([0, 1, 6, 2], dtype=np.uint32)), "coords_batch_idxs contains an index that is out of bounds"),
        (dict(j=np.array([3, 0, 4, 2], dtype=np.uint32)), "params_batch_idxs contains an index that is out of bounds"),
    ]
    for kwargs, message in cases:
        with pytest.raises(RuntimeError) as e:
            call(**kwargs)
        assert str(e.value) == message, (kwargs.keys(), str(e.value))
    du_dx, du_dp, u = call()
    assert du_dx.shape == (4, N, 3) and du_dp.shape == (4, *params.shape) and u.shape == (4,)
    for k in range(4):
        r = impl.execute(coords_batch[ci[k]], params_batch[pi[k]], boxes_batch[ci[k]])
        np.testing.assert_array_equal(du_dx[k], r[0])
        np.testing.assert_array_equal(du_dp[k], r[1])
        assert u[k] == r[2]


def test_context_without_nonbonded_potentials_performs_no_box_check(co, P):
    """tests/test_md.py:946-976: the box-vs-cutoff validation belongs to the nonbonded all-pairs potentials a Context finds among
    its bound potentials; with bonded terms only, a tiny box is accepted.  Plus the set_v_t size message (:118-120)."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s = ts.add_chain_ligand(ts.build_water_box(64, 2.7, seed=3), 8, lamb=0.0)
    bonded = [bp for bp in ts.bound_potentials(s) if not isinstance(bp.potential, (P.Nonbonded, P.NonbondedInteractionGroup))]
    assert 2 <= len(bonded) < len(ts.bound_potentials(s))
    bps = [bp.to_gpu(np.float32).bound_impl for bp in bonded]
    ctxt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.5e-3, 0.0, s.masses, 2022).impl(), bps)
    ctxt.set_box(s.box * 0.01)
    _, boxes = ctxt.multiple_steps(1)
    assert len(boxes) == 1
    with pytest.raises(RuntimeError, match="number of new velocities disagree with current coords"):
        ctxt.set_v_t(np.zeros((s.num_atoms - 1, 3)))


@pytest.mark.parametrize("precision", [np.float32, np.float64])
def test_doubles_from_the_device_equal_the_host_conversions_of_the_raw_accumulators(co, P, precision):
    """execute / execute_batch / execute_batch_sparse / BoundPotential.execute[_batch] return doubles converted on the DEVICE
    (tm_potential_execute_f64 ...).  They must equal, bit for bit, what the reference's binding computes on the host from the raw
    accumulators (wrap_kernels.cpp:1066-1101: FIXED_TO_FLOAT; du_dp_fixed_to_float with the nonbonded columns' 2^36 / 2^37 / 2^38 /
    2^36 and the slices of a SummedPotential; convert_energy_to_fp with NaN for an overflow) -- here: execute_raw + numpy."""
    from timemachine_amd import testsystems as ts

    s = ts.small_solvated_ligand(lamb=0.3)
    state = ts.rbfe_shaped_state(s, 20)
    summed = P.SummedPotential([p for p, _ in state], [q for _, q in state])
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
    impl = summed.to_gpu(precision).unbound_impl
    rng = np.random.default_rng(2)
    xs = np.stack([s.coords + rng.normal(0, 0.002, s.coords.shape) for _ in range(3)])
    boxes = np.stack([s.box] * 3)
    params = np.stack([flat, flat * 1.01])

    def expected(x, prm, box):
        dx, dp, u = impl.execute_raw(x, prm, box, True, True, True)
        e_dx = dx.view(np.int64).astype(np.float64) / 2.0**36
        e_dp = dp.view(np.int64).astype(np.float64) / 2.0**36
        off = 0
        for pot, q in state:  # the nonbonded blocks carry per-column exponents (q, sig, eps, w)
            n = int(np.asarray(q).size)
            if type(pot).__name__.startswith("Nonbonded"):
                blk = dp[off : off + n].view(np.int64).astype(np.float64).reshape(-1, 4)
                e_dp[off : off + n] = (blk / np.array([2.0**36, 2.0**37, 2.0**38, 2.0**36])).reshape(-1)
            off += n
        return e_dx, e_dp, float(u) / 2.0**36

    dx, dp, u = impl.execute(xs[0], flat, boxes[0])
    e = expected(xs[0], flat, boxes[0])
    np.testing.assert_array_equal(dx, e[0])
    np.testing.assert_array_equal(dp, e[1])
    assert u == e[2]
    bdx, bdp, bu = impl.execute_batch(xs, params, boxes, True, True, True)
    assert bdx.shape == (3, 2) + s.coords.shape and bdp.shape == (3, 2, flat.size) and bu.shape == (3, 2)
    ci, pi = np.array([2, 0, 1, 2], dtype=np.uint32), np.array([1, 1, 0, 0], dtype=np.uint32)
    sdx, sdp, su = impl.execute_batch_sparse(xs, params, boxes, ci, pi, True, True, True)
    for i in range(3):
        for j in range(2):
            e = expected(xs[i], params[j], boxes[i])
            np.testing.assert_array_equal(bdx[i, j], e[0])
            np.testing.assert_array_equal(bdp[i, j], e[1])
            assert bu[i, j] == e[2]
    for k in range(4):
        np.testing.assert_array_equal(sdx[k], bdx[ci[k], pi[k]])
        np.testing.assert_array_equal(sdp[k], bdp[ci[k], pi[k]])
        assert su[k] == bu[ci[k], pi[k]]
    # None for what was not asked for, in every form
    assert impl.execute(xs[0], flat, boxes[0], False, False, True)[:2] == (None, None)
    only_u = impl.execute_batch(xs, params, boxes, False, False, True)
    assert only_u[0] is None and only_u[1] is None
    np.testing.assert_array_equal(only_u[2], bu)
    bound = co.BoundPotential(impl, flat)
    b_dx, b_u = bound.execute(xs[1], boxes[1])
    np.testing.assert_array_equal(b_dx, bdx[1, 0])
    assert b_u == bu[1, 0]
    bb_dx, bb_u = bound.execute_batch(xs, boxes, True, True)
    np.testing.assert_array_equal(bb_dx, bdx[:, 0])
    np.testing.assert_array_equal(bb_u, bu[:, 0])
    # an overflowed energy reads NaN (convert_energy_to_fp): three atoms on top of each other -- every clashing pair's energy is
    # clamped to LLONG_MAX (FLOAT_TO_FIXED_ENERGY), and two of those no longer fit the int64 range the conversion accepts
    clash = s.coords.copy()
    clash[3] = clash[0]
    clash[6] = clash[0]
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(precision).unbound_impl
    assert np.isnan(nb.execute(clash, s.nb_params, s.box, False, False, True)[2])
