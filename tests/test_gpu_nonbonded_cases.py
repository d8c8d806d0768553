"""GPU tests (``-m gpu``): cases of the reference's nonbonded / neighbor-list / Hilbert-sort suites not covered elsewhere --
tests/nonbonded/test_nonbonded_all_pairs.py:74-125 (singleton and improper subsets), tests/nonbonded/
test_nonbonded_interaction_group.py:48-75,170-222,301-363 (no interactions, empty / all index sets, the constant-shift
identity against the all-pairs oracle), tests/test_nblist.py:23-25 (empty list), tests/test_hilbert_sort.py (sorted blocks
are compact).  Inputs are synthetic (seeded); the oracle is the checker where a reference value is needed."""
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


def params_with_4d_offsets(rng, params, cutoff):
    """tests/common.py gen_nonbonded_params_with_4d_offsets: w = 0, random in (-cutoff, cutoff), and alternating 0 / cutoff"""
    n = len(params)
    for w in (np.zeros(n), rng.uniform(-cutoff, cutoff, n), cutoff * (np.arange(n) % 2)):
        p = np.array(params)
        p[:, 3] = w
        yield p


def test_nonbonded_all_pairs_singleton_subset(co, P):
    rng = np.random.default_rng(2022)
    num_atoms, beta, cutoff = 231, 2.0, 1.1
    box = 3.0 * np.eye(3)
    conf = rng.uniform(0, 1, size=(num_atoms, 3))
    params = rng.uniform(0, 1, size=(num_atoms, 4))
    for idx in rng.choice(num_atoms, size=(10,)):
        pot = P.NonbondedAllPairs(num_atoms, beta, cutoff, np.array([idx], dtype=np.int32))
        du_dx, du_dp, u = pot.to_gpu(np.float64).unbound_impl.execute(conf, params, box)
        assert (du_dx == 0).all() and (du_dp == 0).all() and u == 0


def test_nonbonded_all_pairs_improper_subset(co, P):
    rng = np.random.default_rng(2023)
    num_atoms, beta, cutoff = 231, 2.0, 1.1
    box = 3.0 * np.eye(3)
    conf = rng.uniform(0, 1, size=(num_atoms, 3))
    params = rng.uniform(0, 1, size=(num_atoms, 4))

    def run(atom_idxs):
        return P.NonbondedAllPairs(num_atoms, beta, cutoff, atom_idxs).to_gpu(np.float64).unbound_impl.execute(conf, params, box)

    a, b = run(None), run(np.arange(num_atoms, dtype=np.int32))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert (np.isnan(a[2]) and np.isnan(b[2])) or a[2] == b[2]


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_nonbonded_interaction_group_zero_interactions(co, P, precision):
    rng = np.random.default_rng(2024)
    num_atoms, num_lig, beta, cutoff = 33, 15, 2.0, 1.1
    box = 10.0 * np.eye(3)
    conf = rng.uniform(0, 1, size=(num_atoms, 3))
    ligand_idxs = rng.choice(num_atoms, size=(num_lig,), replace=False).astype(np.int32)
    conf[ligand_idxs, 0] += 2 * cutoff  # the two groups are out of each other's reach
    params = rng.uniform(0, 1, size=(num_atoms, 4))
    du_dx, du_dp, u = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff).to_gpu(precision).unbound_impl.execute(conf, params, box)
    assert (du_dx == 0).all() and (du_dp == 0).all() and u == 0


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("num_atoms", [50, 231])
def test_nonbonded_interaction_group_empty_or_full_row_set_is_a_no_op(co, P, precision, num_atoms):
    """An interaction group whose row set is empty, or is every atom (so the column set is empty), has nothing to compute:
    zero energy and zero derivatives for every flag combination (the reference supports this for local MD)."""
    rng = np.random.default_rng(num_atoms)
    beta, cutoff = 2.0, 1.1
    box = 3.0 * np.eye(3)
    conf = rng.uniform(0, 3, size=(num_atoms, 3))
    params0 = np.stack([rng.normal(size=num_atoms), rng.uniform(0.05, 0.15, num_atoms), rng.uniform(0.2, 1, num_atoms), np.zeros(num_atoms)], 1)
    gpu = P.NonbondedInteractionGroup(num_atoms, np.array([0], dtype=np.int32), beta, cutoff).to_gpu(precision)
    assert gpu.unbound_impl.execute(conf, params0, box)[2] != 0.0  # as constructed: one row atom against all others
    for ligand_idxs in (np.array([], dtype=np.int32), np.arange(num_atoms, dtype=np.int32)):
        col_atom_idxs = np.setdiff1d(np.arange(num_atoms), ligand_idxs).astype(np.int32)
        gpu.unbound_impl.set_atom_idxs(ligand_idxs, col_atom_idxs)
        for params in params_with_4d_offsets(rng, params0, cutoff):
            for flags in ((True, True, True), (True, False, False), (False, False, True), (False, True, False)):
                du_dx, du_dp, u = gpu.unbound_impl.execute(conf, params, box, *flags)
                assert du_dx is None or (du_dx == 0).all()
                assert du_dp is None or (du_dp == 0).all()
                assert u is None or u == 0


@pytest.mark.parametrize("precision,rtol,atol", [(np.float64, 1e-8, 1e-8), (np.float32, 1e-4, 5e-4)])
@pytest.mark.parametrize("num_atoms_ligand", [1, 15])
@pytest.mark.parametrize("num_atoms", [33, 231])
def test_nonbonded_interaction_group_consistency_allpairs_constant_shift(co, P, precision, rtol, atol, num_atoms_ligand, num_atoms):
    """U(x') - U(x) == U_AB(x') - U_AB(x) when x -> x' translates group A rigidly: the all-pairs side is the oracle (every pair,
    no exclusions), the interaction-group side the GPU."""
    from oracle import ref_potentials as rp

    rng = np.random.default_rng(100 * num_atoms + num_atoms_ligand)
    beta, cutoff = 2.0, 1.1
    box = 3.0 * np.eye(3)
    conf = rng.uniform(0, 3, size=(num_atoms, 3))
    params0 = np.stack([rng.normal(size=num_atoms), rng.uniform(0.05, 0.15, num_atoms), rng.uniform(0.2, 1, num_atoms), np.zeros(num_atoms)], 1)
    ligand_idxs = rng.choice(num_atoms, size=(num_atoms_ligand,), replace=False).astype(np.int32)
    impl = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff).to_gpu(precision).unbound_impl
    conf_prime = np.array(conf)
    conf_prime[ligand_idxs] += rng.normal(0, 0.01, size=(3,))
    for params in params_with_4d_offsets(rng, params0, cutoff):
        ref_delta = rp.nonbonded_all_pairs(conf_prime, params, box, beta, cutoff)[0] - rp.nonbonded_all_pairs(conf, params, box, beta, cutoff)[0]
        test_delta = impl.execute(conf_prime, params, box)[2] - impl.execute(conf, params, box)[2]
        # the differences are of two large numbers: the bar scales with the energies themselves, as the reference's does through rtol on delta + atol
        np.testing.assert_allclose(ref_delta, test_delta, rtol=rtol * 10, atol=atol * max(1.0, abs(impl.execute(conf, params, box)[2])))


def test_empty_neighborlist(co):
    with pytest.raises(RuntimeError, match="Neighborlist N must be at least 1"):
        co.Neighborlist_f32(0)


@pytest.mark.parametrize("block_size", [8, 16, 32])
def test_hilbert_sort_makes_compact_blocks(co, block_size):
    """tests/test_hilbert_sort.py: on a solvated box in input order (molecule by molecule from a lattice, then shuffled like a
    real topology is not), the mean over blocks of the largest intra-block distance drops below 0.6 of the unsorted one."""
    from timemachine_amd import testsystems as ts

    s = ts.build_water_box(2197, 4.05, seed=1)
    rng = np.random.default_rng(0)
    mol_order = rng.permutation(s.num_atoms // 3)
    coords = s.coords.reshape(-1, 3, 3)[mol_order].reshape(-1, 3)  # molecules in arbitrary order, atoms of a molecule together
    L = np.diagonal(s.box)

    def mean_max_block_distance(x):
        out = []
        for b in range(0, len(x), block_size):
            blk = x[b:b + block_size]
            d = blk[:, None, :] - blk[None, :, :]
            d -= L * np.rint(d / L)
            out.append(np.sqrt((d ** 2).sum(-1)).max())
        return np.mean(out)

    perm = co.HilbertSort(len(coords)).sort(coords, s.box)
    assert sorted(perm.tolist()) == list(range(len(coords)))
    assert mean_max_block_distance(coords[perm]) < 0.6 * mean_max_block_distance(coords)
