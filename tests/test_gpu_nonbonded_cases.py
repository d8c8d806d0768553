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


@pytest.mark.parametrize("tiles", [2, 10, 100, 300])
def test_nblist_max_interactions(co, tiles):
    """tests/test_nblist.py:366-383: every atom within the cutoff of every other -- the list fills its worst-case buffers exactly.
    (300 tiles = 9 600 listed atoms per row block: more than the list kernel stages in LDS (NBL_CAND_CAP = 8192), so the cost
    estimates of the chunks past the staging area take their sampled path on atoms re-read from memory.)"""
    rng = np.random.default_rng(2023)
    block_size, cutoff = 32, 10.0
    coords = rng.random(size=(block_size * tiles, 3))
    box = np.eye(3) * 100.0
    nblist = co.Neighborlist_f32(coords.shape[0])
    max_ixn_count = nblist.get_max_ixn_count()
    ixn_list = nblist.get_nblist(coords, box, cutoff)
    assert len(ixn_list) == tiles
    n = len(coords)
    for i, cols in enumerate(ixn_list):
        # upper-triangular list: row block i sees its own block's and every later block's atoms, all of them
        assert sorted(cols) == list(range(i * block_size, n))
    assert nblist.get_tile_ixn_count() * block_size == max_ixn_count


@pytest.mark.parametrize("num_atoms", [35, 129, 1025])
def test_nblist_row_indices_are_order_independent(co, num_atoms):
    """tests/test_nblist.py:188-233: the same row subset in a different order lists the same interactions (as a set; the
    per-block lists follow the order given), in both precisions."""
    from oracle import nblist as onblist

    rng = np.random.default_rng(1234)
    coords = rng.uniform(0, 1, size=(num_atoms, 3)) * (num_atoms / 100.0) ** (1 / 3)  # water-like number density
    box = np.diag(coords.max(0) - coords.min(0) + 0.1 + 2.2)  # > 2 * cutoff in every dimension
    rows = rng.choice(num_atoms, num_atoms // 2, replace=False).astype(np.uint32)
    shuffled = rows.copy()
    rng.shuffle(shuffled)
    assert not np.all(shuffled == rows)
    ref = onblist.brute_force_ixn_list_rows(coords, box, 1.0, rows)
    ref_shuffled = onblist.brute_force_ixn_list_rows(coords, box, 1.0, shuffled)
    all_ref = set(np.concatenate([np.asarray(b, dtype=np.int64) for b in ref]).tolist())
    assert all_ref == set(np.concatenate([np.asarray(b, dtype=np.int64) for b in ref_shuffled]).tolist())
    for cls in (co.Neighborlist_f32, co.Neighborlist_f64):
        nb = cls(num_atoms)
        for idxs, expect in ((rows, ref), (shuffled, ref_shuffled)):
            nb.set_row_idxs(idxs)
            test = nb.get_nblist(coords, box, 1.0)
            assert len(test) == len(expect)
            for a, b in zip(expect, test):
                assert sorted(a) == sorted(b)


@pytest.mark.parametrize("cutoff", [1.0, 1.2])
def test_nblist_density_of_a_dhfr_sized_box(co, cutoff):
    """tests/test_nblist.py:459-472: in Hilbert order the mean fraction of a listed 32 x 32 tile's slots that are inside the
    cutoff is above 10 % (DHFR-sized synthetic water box; measured here: about 30 %), and well above the unsorted value."""
    from timemachine_amd import testsystems as ts

    s = ts.dhfr_sized_water_box()
    rng = np.random.default_rng(5)
    mol_order = rng.permutation(s.num_atoms // 3)
    coords = s.coords.reshape(-1, 3, 3)[mol_order].reshape(-1, 3)
    L = np.diagonal(s.box)
    nblist = co.Neighborlist_f32(len(coords))

    def mean_tile_density(x):
        lists = nblist.get_nblist(x, s.box, cutoff)
        dens = []
        for i in list(range(0, len(lists), 37)):  # a sample of the row blocks keeps the numpy side in seconds
            cols = np.asarray(lists[i], dtype=np.int64)
            rows = x[i * 32:(i + 1) * 32]
            for cb in np.unique(cols // 32):
                c = cols[cols // 32 == cb]
                d = rows[:, None, :] - x[c][None, :, :]
                d -= L * np.rint(d / L)
                dens.append(((d ** 2).sum(-1) < cutoff * cutoff).sum() / 1024.0)
        return float(np.mean(dens))

    unsorted = mean_tile_density(coords)
    perm = co.HilbertSort(len(coords)).sort(coords, s.box)
    density = mean_tile_density(coords[perm])
    assert density > 0.10 and density > 2.0 * unsorted, (density, unsorted)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_nblist_rebuild(co, P, precision):
    """tests/nonbonded/test_nonbonded.py:67-116: a potential that keeps its padded list (displacements stay within padding / 2,
    so no rebuild is triggered) gives the same bits as one with padding 0, which lists afresh on every call -- through
    execute and through execute_du_dx.  (The reference passes this in f64 only; integer accumulation makes it hold in f32 too.)"""
    from timemachine_amd import testsystems as ts

    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    rng = np.random.default_rng(2021)
    padding = 0.1
    ref = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, nblist_padding=0.0).to_gpu(precision).unbound_impl
    test = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, nblist_padding=padding).to_gpu(precision).unbound_impl
    deltas = (rng.random((N, 3)) - 0.5) / (0.5 * (2 * np.sqrt(3)) / padding)  # |delta| < padding / 2: no rebuild
    assert np.all(np.linalg.norm(deltas, axis=1) < padding / 2)
    for x in (s.coords, s.coords + deltas):
        a = ref.execute(x, s.nb_params, s.box)
        b = test.execute(x, s.nb_params, s.box)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2]
        np.testing.assert_array_equal(ref.execute_du_dx(x, s.nb_params, s.box), test.execute_du_dx(x, s.nb_params, s.box))


def _ig_system(rng, num_atoms, box_len=3.0):
    conf = rng.uniform(0, box_len, size=(num_atoms, 3))
    params = np.stack([rng.normal(size=num_atoms), rng.uniform(0.05, 0.15, num_atoms), rng.uniform(0.2, 1, num_atoms), np.zeros(num_atoms)], 1)
    return conf, params, box_len * np.eye(3)


@pytest.mark.parametrize("precision,rtol,atol", [(np.float64, 1e-8, 1e-8), (np.float32, 1e-4, 5e-4)])
@pytest.mark.parametrize("num_atoms,num_atoms_ligand", [(50, 1), (50, 15), (231, 15)])
@pytest.mark.parametrize("num_col_atoms", [1, 10, 33])
def test_nonbonded_interaction_group_neighborlist_rebuild(co, P, precision, rtol, atol, num_atoms, num_atoms_ligand, num_col_atoms):
    """tests/nonbonded/test_nonbonded_interaction_group.py:121-166: moving the column atoms far enough triggers a list rebuild;
    energies and forces agree with the oracle before and after, for every 4D-offset pattern."""
    from oracle import ref_potentials as rp

    rng = np.random.default_rng(7 * num_atoms + num_atoms_ligand + 100 * num_col_atoms)
    beta, cutoff = 2.0, 1.1
    conf, params0, box = _ig_system(rng, num_atoms)
    ligand_idxs = rng.choice(num_atoms, size=(num_atoms_ligand,), replace=False).astype(np.int32)
    host_idxs = np.setdiff1d(np.arange(num_atoms), ligand_idxs).astype(np.int32)
    col_atom_idxs = rng.choice(host_idxs, size=(num_col_atoms,), replace=False).astype(np.int32)
    impl = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff, col_atom_idxs=col_atom_idxs).to_gpu(precision).unbound_impl

    def check(x, prm):
        u_ref, du_dx_ref, du_dp_ref = rp.value_and_grads(
            lambda xx, pp: rp.nonbonded_interaction_group_energy(xx, pp, rp._t(box), ligand_idxs, col_atom_idxs, beta, cutoff), x, prm)
        du_dx, du_dp, u = impl.execute(x, prm, box)
        np.testing.assert_allclose(u, u_ref, rtol=rtol, atol=atol)
        scale = max(1.0, np.abs(du_dx_ref).max())
        np.testing.assert_allclose(du_dx, du_dx_ref, rtol=rtol * 10, atol=atol * scale)
        np.testing.assert_allclose(du_dp, du_dp_ref, rtol=rtol * 100, atol=atol * 10 * max(1.0, np.abs(du_dp_ref).max()))

    for params in params_with_4d_offsets(rng, params0, cutoff):
        check(conf, params)
        conf[col_atom_idxs] += rng.random(size=(len(col_atom_idxs), 3)) * (cutoff**2)
        check(conf, params)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("num_atoms,num_atoms_ligand", [(33, 1), (33, 15), (231, 15)])
def test_nonbonded_interaction_group_set_atom_idxs(co, P, precision, num_atoms, num_atoms_ligand):
    """tests/nonbonded/test_nonbonded_interaction_group.py:365-413: set_atom_idxs makes the object the same, bit for bit, as a
    fresh one with those groups, and setting the original groups again (in another order) restores the original bits."""
    rng = np.random.default_rng(31 * num_atoms + num_atoms_ligand)
    beta, cutoff = 2.0, 1.1
    box = 3.0 * np.eye(3)
    conf = rng.uniform(0, cutoff * 10, size=(num_atoms, 3))
    params = rng.uniform(0, 1, size=(num_atoms, 4))
    ligand_idxs = rng.choice(num_atoms, size=(num_atoms_ligand,), replace=False).astype(np.int32)
    other_idxs = np.setdiff1d(np.arange(num_atoms), ligand_idxs)
    secondary = rng.choice(other_idxs, size=(1,), replace=False).astype(np.int32)
    pot = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff).to_gpu(precision).unbound_impl
    ref = pot.execute(conf, params, box)
    pot.set_atom_idxs(secondary, np.setdiff1d(np.arange(num_atoms), secondary).astype(np.int32))
    diff = pot.execute(conf, params, box)
    assert np.any(diff[0] != ref[0]) and np.any(diff[1] != ref[1]) and not np.allclose(ref[2], diff[2], equal_nan=False)
    fresh = P.NonbondedInteractionGroup(num_atoms, secondary, beta, cutoff).to_gpu(precision).unbound_impl.execute(conf, params, box)
    np.testing.assert_array_equal(fresh[0], diff[0])
    np.testing.assert_array_equal(fresh[1], diff[1])
    np.testing.assert_equal(fresh[2], diff[2])  # NaN (overflowed energy) equals NaN here, as in the reference
    rng.shuffle(ligand_idxs)
    pot.set_atom_idxs(ligand_idxs, np.setdiff1d(np.arange(num_atoms), ligand_idxs).astype(np.int32))
    again = pot.execute(conf, params, box)
    np.testing.assert_array_equal(again[0], ref[0])
    np.testing.assert_array_equal(again[1], ref[1])
    np.testing.assert_equal(again[2], ref[2])


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("num_atoms,num_atoms_ligand", [(33, 1), (33, 15), (231, 1), (231, 15)])
def test_nonbonded_ixn_group_order_independent(co, P, precision, num_atoms, num_atoms_ligand):
    """tests/nonbonded/test_nonbonded_interaction_group.py:417-449: with and without the Hilbert sort, bit for bit"""
    rng = np.random.default_rng(17 * num_atoms + num_atoms_ligand)
    beta, cutoff = 2.0, 1.1
    conf, params0, box = _ig_system(rng, num_atoms)
    ligand_idxs = rng.choice(num_atoms, size=(num_atoms_ligand,), replace=False).astype(np.int32)
    a = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff).to_gpu(precision).unbound_impl
    b = P.NonbondedInteractionGroup(num_atoms, ligand_idxs, beta, cutoff, disable_hilbert_sort=True).to_gpu(precision).unbound_impl
    for params in params_with_4d_offsets(rng, params0, cutoff):
        ra, rb = a.execute(conf, params, box), b.execute(conf, params, box)
        np.testing.assert_array_equal(ra[0], rb[0])
        np.testing.assert_array_equal(ra[1], rb[1])
        assert ra[2] == rb[2]


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("select_atom_indices", [False, True])
@pytest.mark.parametrize("num_atoms", [33, 65, 231, 1050])
def test_nonbonded_correctness_on_prefixes_and_subsets(co, P, precision, select_atom_indices, num_atoms):
    """tests/nonbonded/test_nonbonded.py:124-156: the first num_atoms atoms of a solvated system with the exclusions that lie
    inside the prefix, optionally restricted to a random half of the atoms (atom_idxs) -- u, du_dx, du_dp against the oracle,
    all eight flag combinations, each executed twice and compared bitwise (compare_forces)."""
    from oracle import ref_potentials as rp
    from test_gpu_parity import compare_forces
    from timemachine_amd import testsystems as ts

    s = ts.small_solvated_ligand(lamb=0.3)
    rng = np.random.default_rng(num_atoms)
    x = s.coords[:num_atoms].astype(np.float32).astype(np.float64)
    params = s.nb_params[:num_atoms]
    keep = np.all(s.exclusion_idxs < num_atoms, axis=1)
    excl, scales = np.ascontiguousarray(s.exclusion_idxs[keep]), np.ascontiguousarray(s.scale_factors[keep])
    atom_idxs = np.array(rng.choice(num_atoms, num_atoms // 2, replace=False), dtype=np.int32) if select_atom_indices else None
    pot = P.Nonbonded(num_atoms, excl, scales, s.beta, s.cutoff, atom_idxs=atom_idxs)
    u, du_dx, du_dp = rp.nonbonded(x, params, s.box, excl, scales, s.beta, s.cutoff, atom_idxs=atom_idxs)
    compare_forces(pot.to_gpu(precision).unbound_impl, x, params, s.box, float(u), du_dx, du_dp, precision)


def test_pair_list_constructor_validation(co, P):
    """tests/nonbonded/test_nonbonded_pair_list.py:10-24 and test_nonbonded_precomputed.py:10-19"""
    with pytest.raises(RuntimeError) as e:
        P.NonbondedPairList([0], [0], 2.0, 1.1).to_gpu(np.float32).unbound_impl
    assert "pair_idxs.size() must be even, but got 1" in str(e)
    with pytest.raises(RuntimeError) as e:
        P.NonbondedPairList([(0, 0)], [(1, 1)], 2.0, 1.1).to_gpu(np.float32).unbound_impl
    assert "illegal pair with src == dst: 0, 0" in str(e)
    with pytest.raises(RuntimeError) as e:
        P.NonbondedPairList([(0, 1)], [(1, 1), (2, 2)], 2.0, 1.1).to_gpu(np.float32).unbound_impl
    assert "expected same number of pairs and scale tuples, but got 1 != 2" in str(e)
    with pytest.raises(RuntimeError) as e:
        P.NonbondedPairListPrecomputed([0], 2.0, 1.1).to_gpu(np.float32).unbound_impl
    assert "idxs.size() must be exactly 2*B" in str(e)
    with pytest.raises(RuntimeError) as e:
        P.NonbondedPairListPrecomputed([(0, 0)], 2.0, 1.1).to_gpu(np.float32).unbound_impl
    assert "illegal pair with src == dst: 0, 0" in str(e)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("ixn_group_size", [2, 33, 231])
def test_nonbonded_pair_list_correctness(co, P, precision, ixn_group_size):
    """tests/nonbonded/test_nonbonded_pair_list.py:31-60: all pairs between two disjoint random groups, random rescale masks,
    the three 4D-offset patterns -- against the oracle, all flag combinations (compare_forces)."""
    from oracle import ref_potentials as rp
    from test_gpu_parity import compare_forces
    from timemachine_amd import testsystems as ts

    s = ts.small_solvated_ligand(lamb=0.0)
    rng = np.random.default_rng(ixn_group_size)
    x = s.coords.astype(np.float32).astype(np.float64)
    beta, cutoff = 2.0, 1.1
    atom_idxs = rng.choice(s.num_atoms, size=(2, ixn_group_size), replace=False).astype(np.int32)
    pair_idxs = np.ascontiguousarray(np.stack(np.meshgrid(atom_idxs[0], atom_idxs[1])).reshape(2, -1).T.astype(np.int32))
    rescale = rng.uniform(0, 1, size=(len(pair_idxs), 2))
    impl = P.NonbondedPairList(pair_idxs, rescale, beta, cutoff).to_gpu(precision).unbound_impl
    for params in params_with_4d_offsets(rng, s.nb_params, cutoff):
        u, du_dx, du_dp = rp.nonbonded_pair_list(x, params, s.box, pair_idxs, rescale, beta, cutoff)
        compare_forces(impl, x, params, s.box, float(u), du_dx, du_dp, precision)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("cutoff", [1.1, 10000.0])
@pytest.mark.parametrize("ixn_group_size", [4, 33, 231])
def test_nonbonded_pair_list_precomputed_correctness(co, P, precision, cutoff, ixn_group_size):
    """tests/nonbonded/test_nonbonded_precomputed.py:27-60: random pairs of a 25 358-atom configuration with per-pair
    parameters (q_ij, sig_ij, eps_ij, w offset >= 0), a box that the potential must ignore... it does not: the reference's
    kernel applies the minimum image like every other nonbonded term; finite and infinite cutoff."""
    from oracle import ref_potentials as rp
    from test_gpu_parity import compare_forces

    num_atoms = 25358
    rng = np.random.default_rng(1000 + ixn_group_size)
    pair_idxs = np.array([rng.choice(num_atoms, 2, replace=False) for _ in range(ixn_group_size)], dtype=np.int32)
    params0 = rng.uniform(0, 1, size=(ixn_group_size, 4))
    params0[:, 0] -= 0.5
    params0[:, 1] /= 5
    conf = (rng.uniform(0, 1, size=(num_atoms, 3)) * 3).astype(np.float32).astype(np.float64)
    box = np.diag(1 + rng.uniform(0, 1, size=3) * 3)
    impl = P.NonbondedPairListPrecomputed(pair_idxs, 2.0, cutoff).to_gpu(precision).unbound_impl
    wcut = min(cutoff, 2.0)
    for w in (np.zeros(ixn_group_size), rng.uniform(0, wcut, ixn_group_size), wcut * (np.arange(ixn_group_size) % 2)):
        params = params0.copy()
        params[:, 3] = w
        u, du_dx, du_dp = rp.nonbonded_pair_list_precomputed(conf, params, box, pair_idxs, 2.0, cutoff)
        compare_forces(impl, conf, params, box, float(u), du_dx, du_dp, precision)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_static_complete_list_of_small_systems_changes_no_bit(co, P, precision):
    """Potentials over few atoms keep a static, complete interaction list (every column block listed for every row block: no
    displacement can invalidate it, no list kernel runs on MD steps; EXPERIMENTS.md, History, item 11).  The exact test d2 < cutoff^2 of the
    tile kernel decides alone either way, so forces, energies, du/dp and whole trajectories -- through Hilbert re-sorts, a
    barostat and host-API calls in between -- must be bit-identical to the listed pipeline; and the static one never rebuilds."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    x0 = s.coords.astype(np.float32).astype(np.float64)

    def run(static_max_k):
        before = co.debug_set_static_list_max_k(static_max_k)
        try:
            nb = P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(precision).unbound_impl
            raw = nb.execute_raw(x0, s.nb_params, s.box)
            bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision)]
            baro = MonteCarloBarostat(N, 1.0, 300.0, ts.molecule_groups(s), 15, 3).impl(bps)
            ctxt = co.Context(x0, np.zeros_like(x0), s.box, LangevinIntegrator(300.0, 1.0e-3, 5.0, s.masses, 11).impl(), bps, movers=[baro])
            xs, boxes = ctxt.multiple_steps(150, 25)
            u_mid = bps[-1].execute(ctxt.get_x_t(), ctxt.get_box(), False, True)[1]  # a host-API call between MD calls
            xs2, boxes2 = ctxt.multiple_steps(130, 65)  # crosses the re-sort at 200 calls
            all_pairs = bps[-1].get_potential().get_potentials()[0]
            return raw, xs, boxes, u_mid, xs2, boxes2, all_pairs.get_build_count(), baro.get_counters()
        finally:
            co.debug_set_static_list_max_k(before)

    a = run(0)
    b = run(1 << 20)
    for k in (0, 1):
        np.testing.assert_array_equal(a[0][k], b[0][k])
    assert a[0][2] == b[0][2]
    for k in (1, 2, 4, 5):
        np.testing.assert_array_equal(a[k], b[k])
    assert a[3] == b[3] and a[7] == b[7] and a[7][0] > 0
    assert b[6] < a[6] and b[6] <= 8, (a[6], b[6])  # listed: a rebuild every few steps; static: only with a new order


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("static_list", [True, False])
@pytest.mark.parametrize("case", ["dense_alchemical", "sparse_large_box", "drifted_images", "tiny_box"])
def test_tile_filter_paths_against_the_oracle(co, P, precision, case, static_list):
    """The tile kernel's phase-1 filter has a fast path (Gram form, no w term: flat items off the list's diagonal) and general
    ones (per-atom w; items whose extent exceeds the Gram form's bound or half a box length: explicit differences with the
    minimum image; diagonal tiles with the row < col test), chosen per item.  Whatever the filter does, the exact test in
    phase 2 decides -- so every combination must reproduce the oracle: a dense box with per-atom w (all items non-flat), a
    sparse 60 nm box (row blocks span many nm: no Gram form), coordinates drifted by whole box lengths per atom (wrapped on the
    way to the tile origin: phase 2 re-images), and a box barely twice the cutoff (hardly any item is compact) -- each on the
    static complete list these sizes get by default (items then pair every row block with every column block, however far
    apart) and on a built neighbor list."""
    from oracle import ref_potentials as rp
    from test_gpu_parity import compare_forces

    rng = np.random.default_rng({"dense_alchemical": 1, "sparse_large_box": 2, "drifted_images": 3, "tiny_box": 4}[case])
    cutoff, beta = 1.2, 2.0
    if case == "sparse_large_box":
        n, edge = 700, 60.0
        # clusters, so that pairs inside the cutoff exist at all at this density
        centres = rng.uniform(0.0, edge, (35, 3))
        x = (centres[rng.integers(0, 35, n)] + rng.normal(0.0, 0.45, (n, 3))) % edge
    elif case == "tiny_box":
        n, edge = 500, 2.45
        x = rng.uniform(0.0, edge, (n, 3))
    else:
        n, edge = 900, 3.0
        x = rng.uniform(0.0, edge, (n, 3))
    # keep clear of clashes: the oracle and the kernels agree there too, but tolerances are relative to huge numbers then
    box = np.eye(3) * edge
    for _ in range(50):
        d = x[:, None, :] - x[None, :, :]
        d -= edge * np.round(d / edge)
        r = np.sqrt((d * d).sum(-1)) + np.eye(n) * 10.0
        i, j = np.nonzero(r < 0.17)
        if len(i) == 0:
            break
        x[i] += rng.normal(0.0, 0.1, (len(i), 3))
        x %= edge
    if case == "drifted_images":
        # (f32: one box length -- at +-9 nm the f32 coordinates themselves carry 1e-6 nm, 1.2e-4 of a steep LJ force)
        reach = 3 if precision == np.float64 else 1
        x = x + edge * rng.integers(-reach, reach + 1, (n, 3))
    x = x.astype(np.float32).astype(np.float64)
    params = np.zeros((n, 4))
    params[:, 0] = rng.normal(0.0, 0.4, n)
    params[:, 1] = rng.uniform(0.05, 0.17, n)
    params[:, 2] = np.where(rng.random(n) < 0.4, rng.uniform(0.2, 1.0, n), 0.0)
    if case in ("dense_alchemical", "tiny_box"):
        params[:, 3] = np.where(rng.random(n) < 0.5, rng.uniform(0.0, cutoff, n), 0.0)
    params = params.astype(np.float32).astype(np.float64)
    excl = np.zeros((0, 2), dtype=np.int32)
    scales = np.zeros((0, 2))
    pot = P.Nonbonded(n, excl, scales, beta, cutoff)
    u, du_dx, du_dp = rp.nonbonded(x, params, box, excl, scales, beta, cutoff)
    previous = co.debug_set_static_list_max_k(4608 if static_list else 0)
    try:
        compare_forces(pot.to_gpu(precision).unbound_impl, x, params, box, float(u), du_dx, du_dp, precision)
    finally:
        co.debug_set_static_list_max_k(previous)
