"""GPU tests (``-m gpu``) of local MD (Context.setup_local_md / multiple_steps_local / multiple_steps_local_selection) after
the reference's tests/test_md.py: validation (:250-317, :319-387), consistency (:390-506), the entire system (:545-583), no
free particles (:586-627), initialization (:630-710), selection masks (:713-790) -- on synthetic systems
(timemachine_amd.testsystems), plus what only a pinned oracle allows: the reference atom and the free set a call picks are
predicted atom by atom by oracle/local_md.py (same mt19937 draw; the build's own Philox uniforms)."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TEMP = 300.0  # timemachine/constants.py DEFAULT_TEMP


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


def tiny_nb_system(seed, N=8, cutoff=1.0, scale=2.0):
    """N random atoms with two exclusions, as prepare_nb_system(coords, E=2, p_scale=3.0, cutoff=1.0) of the reference tests"""
    from timemachine_amd import potentials as P

    rng = np.random.default_rng(seed)
    coords = rng.random((N, 3)) * scale
    params = np.stack([(rng.random(N) - 0.5) * 3.0 * np.sqrt(138.935456), rng.random(N) * 0.15 + 0.1, rng.random(N) * 0.8 + 0.2, np.zeros(N)], axis=1)
    excl = np.array([[0, 1], [2, 3]], dtype=np.int32)
    scales = np.ones((2, 2))
    pot = P.Nonbonded(N, excl, scales, 2.0, cutoff)
    masses = rng.random(N) + 0.5
    return coords, params, pot, masses


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_multiple_steps_local_validation(co, freeze_reference):
    from timemachine_amd.lib import VelocityVerletIntegrator

    coords, params, pot, masses = tiny_nb_system(2022)
    N = len(coords)
    box = np.eye(3) * 3.0
    v0 = np.zeros_like(coords)
    bps = [pot.bind(params).to_gpu(np.float32).bound_impl]
    verlet = VelocityVerletIntegrator(1.5e-3, masses)

    ctxt = co.Context(coords, v0, box, verlet.impl(), bps)
    # without an explicit setup the temperature comes from the integrator: only a Langevin integrator has one
    with pytest.raises(RuntimeError, match="integrator must be LangevinIntegrator."):
        ctxt.multiple_steps_local(100, np.array([0], dtype=np.int32))

    ctxt = co.Context(coords, v0, box, verlet.impl(), bps)
    with pytest.raises(RuntimeError, match="temperature must be greater than 0"):
        ctxt.setup_local_md(0.0, freeze_reference)
    ctxt.setup_local_md(TEMP, freeze_reference)
    radius = 1.2
    with pytest.raises(RuntimeError, match="indices can't be empty"):
        ctxt.multiple_steps_local(100, np.array([], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="index values must be less than N"):
        ctxt.multiple_steps_local(100, np.array([N * 2], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="index values must be greater or equal to zero"):
        ctxt.multiple_steps_local(100, np.array([-1], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="atom indices must be unique"):
        ctxt.multiple_steps_local(100, np.array([1, 1], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="radius must be greater or equal to 0.1"):
        ctxt.multiple_steps_local(100, np.array([1], dtype=np.int32), radius=0.01)
    with pytest.raises(RuntimeError, match="k must be at least one"):
        ctxt.multiple_steps_local(100, np.array([1], dtype=np.int32), k=0.0)
    with pytest.raises(RuntimeError, match=re.escape("k must be less than than 1e+06")):
        ctxt.multiple_steps_local(100, np.array([1], dtype=np.int32), k=1e7)
    with pytest.raises(RuntimeError, match="store_x_interval must be greater than or equal to zero"):
        ctxt.multiple_steps_local(100, np.array([1], dtype=np.int32), store_x_interval=-1)
    with pytest.raises(RuntimeError, match="local steps must be at least one"):
        ctxt.multiple_steps_local(0, np.array([1], dtype=np.int32))
    with pytest.raises(TypeError):
        ctxt.multiple_steps_local(100, np.array([1], dtype=np.int64))
    # and a valid call still works after all the refused ones (NVE local MD with an explicit temperature)
    xs, boxes = ctxt.multiple_steps_local(10, np.array([1], dtype=np.int32), store_x_interval=5)
    assert xs.shape == (2, N, 3) and boxes.shape == (2, 3, 3)


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_multiple_steps_local_selection_validation(co, freeze_reference):
    from timemachine_amd.lib import VelocityVerletIntegrator

    coords, params, pot, masses = tiny_nb_system(2022)
    N = len(coords)
    box = np.eye(3) * 3.0
    v0 = np.zeros_like(coords)
    bps = [pot.bind(params).to_gpu(np.float32).bound_impl]
    # a selection does not depend on a temperature: compatible with local NVE
    ctxt = co.Context(coords, v0, box, VelocityVerletIntegrator(1.5e-3, masses).impl(), bps)
    ctxt.setup_local_md(TEMP, freeze_reference)
    ref, radius = 0, 1.2
    with pytest.raises(RuntimeError, match="indices can't be empty"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="index values must be less than N"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([N * 2], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="index values must be greater or equal to zero"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([-1], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="atom indices must be unique"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([1, 1], dtype=np.int32), radius=radius)
    with pytest.raises(RuntimeError, match="radius must be greater or equal to 0.1"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([1], dtype=np.int32), radius=0.01)
    with pytest.raises(RuntimeError, match="k must be at least one"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([1], dtype=np.int32), k=0.0)
    with pytest.raises(RuntimeError, match=re.escape("k must be less than than 1e+06")):
        ctxt.multiple_steps_local_selection(100, ref, np.array([1], dtype=np.int32), k=1e7)
    with pytest.raises(RuntimeError, match="reference idx must not be in selection idxs"):
        ctxt.multiple_steps_local_selection(100, ref, np.array([ref], dtype=np.int32))
    with pytest.raises(RuntimeError, match=f"reference idx must be at least 0 and less than {N}"):
        ctxt.multiple_steps_local_selection(100, N, np.array([3], dtype=np.int32))
    with pytest.raises(RuntimeError, match=f"reference idx must be at least 0 and less than {N}"):
        ctxt.multiple_steps_local_selection(100, -1, np.array([3], dtype=np.int32))
    with pytest.raises(RuntimeError, match="store_x_interval must be greater than or equal to zero"):
        ctxt.multiple_steps_local_selection(100, 1, np.array([2], dtype=np.int32), store_x_interval=-1)
    x0 = ctxt.get_x_t()
    xs, _ = ctxt.multiple_steps_local_selection(20, 1, np.array([2, 5], dtype=np.int32))
    moved = np.flatnonzero(np.any(xs[-1] != x0, axis=1))
    np.testing.assert_array_equal(moved, [2, 5] if freeze_reference else [1, 2, 5])


@pytest.fixture(scope="module")
def solvated():
    """a 30-atom ligand in a 4.0 nm water box (~6.4k atoms), relaxed for a moment at constant volume"""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops

    custom_ops.set_device(0)
    s = ts.config4_solvated_ligand()
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    ctxt = custom_ops.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(TEMP, 1.0e-3, 10.0, s.masses, 1).impl(), bps)
    ctxt.multiple_steps(1000, 0)
    return s, ctxt.get_x_t()


def make_bps(s, precision=np.float32):
    from timemachine_amd import testsystems as ts

    return [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s)]


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_multiple_steps_local_consistency(co, solvated, freeze_reference):
    """near the local idxs atoms move, far away nothing does; the context's potentials come back bitwise unchanged; a
    barostat in the context does not run; wrapping everything into one SummedPotential gives identical frames"""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s, coords = solvated
    N = s.num_atoms
    box, masses = s.box, s.masses
    v0 = np.zeros_like(coords)
    dt, friction, seed, radius, num_steps, x_interval = 1.5e-3, 0.0, 2022, 1.2, 500, 100
    bps = make_bps(s)
    reference_values = [bp.execute(coords, box) for bp in bps]
    # the first five atoms of the ligand chain: a compact group (the reference's test uses biphenyl), so that whichever of
    # them becomes the reference atom, the others are well inside the radius and certainly free
    local_idxs = np.arange(N - 30, N - 25, dtype=np.int32)
    intg = LangevinIntegrator(TEMP, dt, friction, masses, seed)

    ctxt = co.Context(coords, v0, box, intg.impl(), bps)
    ctxt.setup_local_md(TEMP, freeze_reference)
    xs, boxes = ctxt.multiple_steps_local(num_steps, local_idxs, store_x_interval=x_interval, radius=radius)
    assert xs.shape[0] == num_steps // x_interval and boxes.shape[0] == num_steps // x_interval
    for b in boxes:
        np.testing.assert_array_equal(b, box)

    expected_to_move = len(local_idxs) - 1 if freeze_reference else len(local_idxs)
    assert np.all(coords[local_idxs] != xs[-1][local_idxs], axis=1).sum() == expected_to_move

    # whoever is within the radius (+ a margin for the probabilistic shell) of the local atoms may have moved; nobody else has
    nblist = co.Neighborlist_f32(N)
    nblist.set_row_idxs(local_idxs.astype(np.uint32))
    # (k/4 0.5^4 = 156 kJ/mol = 63 kT: nobody further out than that is ever selected)
    near = np.concatenate(nblist.get_nblist(coords, box, radius + 0.5))
    moving_idxs = np.unique(np.concatenate([local_idxs, near.reshape(-1)])).astype(np.int64)
    assert np.any(coords[moving_idxs] != xs[-1][moving_idxs])
    frozen_idxs = np.delete(np.arange(N), moving_idxs)
    assert len(frozen_idxs) > 0
    np.testing.assert_array_equal(coords[frozen_idxs], xs[-1][frozen_idxs])
    # the diagnostic agrees with the frames: the atoms reported free are exactly the atoms that moved
    ref_atom, free = ctxt.local_md_last_selection()
    assert ref_atom in local_idxs
    np.testing.assert_array_equal(np.flatnonzero(np.any(xs[-1] != coords, axis=1)), free)
    assert (ref_atom in free) == (not freeze_reference)
    assert np.all(np.isfinite(xs))

    # local MD narrows the nonbonded potential while it runs: afterwards every potential answers as before, bit for bit
    for (ref_du_dx, ref_u), bp in zip(reference_values, bps):
        du_dx, u = bp.execute(coords, box)
        np.testing.assert_array_equal(ref_du_dx, du_dx)
        np.testing.assert_equal(ref_u, u)

    # a barostat in the context stays out of local MD
    groups = [list(range(3 * i, 3 * i + 3)) for i in range((N - 30) // 3)] + [list(range(N - 30, N))]
    baro = MonteCarloBarostat(N, 1.0, TEMP, groups, 1, seed).impl(bps)
    ctxt = co.Context(coords, v0, box, intg.impl(), bps, movers=[baro])
    ctxt.setup_local_md(TEMP, freeze_reference)
    baro_xs, baro_boxes = ctxt.multiple_steps_local(num_steps, local_idxs, store_x_interval=x_interval, radius=radius)
    np.testing.assert_array_equal(baro_xs, xs)
    np.testing.assert_array_equal(baro_boxes, boxes)

    # one SummedPotential around everything: the all-pairs potential is found inside (Summed -> Fanout -> AllPairs)
    ubps = ts.bound_potentials(s)
    summed = P.SummedPotential([bp.potential for bp in ubps], [bp.params for bp in ubps])
    flat = np.concatenate([np.asarray(bp.params).reshape(-1) for bp in ubps])
    bp = summed.bind(flat).to_gpu(np.float32).bound_impl
    ctxt = co.Context(coords, v0, box, intg.impl(), [bp])
    ctxt.setup_local_md(TEMP, freeze_reference)
    summed_xs, summed_boxes = ctxt.multiple_steps_local(num_steps, local_idxs, store_x_interval=x_interval, radius=radius)
    np.testing.assert_array_equal(summed_xs, xs)
    np.testing.assert_array_equal(summed_boxes, boxes)


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_multiple_steps_local_entire_system(co, solvated, freeze_reference):
    """radius = inf selects everything: with a frozen reference exactly one atom stands still, otherwise none"""
    from timemachine_amd.lib import LangevinIntegrator

    s, coords = solvated
    N = s.num_atoms
    local_idxs = np.arange(N - 30, N, dtype=np.int32)
    ctxt = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 2022).impl(), make_bps(s))
    ctxt.setup_local_md(TEMP, freeze_reference)
    xs, boxes = ctxt.multiple_steps_local(5, local_idxs, radius=np.inf)
    assert xs.shape[0] == 1
    assert np.all(np.isfinite(xs))
    if freeze_reference:
        assert np.all(xs[0] == coords, axis=1).sum() == 1, "Expected only a single atom to be stationary"
    else:
        assert np.all(xs[0] != coords), "All coordinates should have moved"
    # and the steps are the steps global MD would have taken for the atoms that move (same forces, same noise, same update)
    glob = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 2022).impl(), make_bps(s))
    gx, _ = glob.multiple_steps(1)
    one = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 2022).impl(), make_bps(s))
    one.setup_local_md(TEMP, freeze_reference)
    lx, _ = one.multiple_steps_local(1, local_idxs, radius=np.inf)
    ref_atom, free = one.local_md_last_selection()
    if freeze_reference:
        np.testing.assert_array_equal(np.delete(lx[0], ref_atom, axis=0), np.delete(gx[0], ref_atom, axis=0))
    else:
        np.testing.assert_array_equal(lx[0], gx[0])


def test_multiple_steps_local_no_free_particles(co):
    """a reference atom far from everything, the smallest radius, the weakest restraint: nobody is selected"""
    from timemachine_amd.lib import LangevinIntegrator

    seed, N = 2023, 100
    coords, params, pot, masses = tiny_nb_system(seed, N=N)
    rng = np.random.default_rng(seed)
    v0 = rng.uniform(0.0, 1.0, size=coords.shape)
    local_idxs = np.array([N - 1], dtype=np.int32)
    coords[local_idxs] += 100.0
    bps = [pot.bind(params).to_gpu(np.float32).bound_impl]
    ctxt = co.Context(coords, v0, np.eye(3) * 1000.0, LangevinIntegrator(TEMP, 1.5e-3, 0.0, masses, seed).impl(), bps)
    with pytest.raises(RuntimeError, match="no free particles"):
        ctxt.multiple_steps_local(1, local_idxs, radius=0.1, k=1.0, seed=seed)
    # the failed setup leaves the context usable for global MD, and the potential as it was
    du_dx_before, u_before = bps[0].execute(coords, np.eye(3) * 1000.0)
    ctxt.multiple_steps(2)
    ctxt.set_x_t(coords)
    du_dx_after, u_after = bps[0].execute(coords, np.eye(3) * 1000.0)
    np.testing.assert_array_equal(du_dx_before, du_dx_after)
    np.testing.assert_equal(u_before, u_after)


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_local_md_initialization(co, freeze_reference):
    """setting local MD up changes neither global MD nor local MD"""
    from timemachine_amd.lib import LangevinIntegrator

    seed = 2023
    coords, params, pot, masses = tiny_nb_system(seed)
    box = np.eye(3) * 3.0
    v0 = np.zeros_like(coords)
    local_idxs = np.array([len(coords) - 1], dtype=np.int32)
    nb = pot.to_gpu(np.float32)
    bps = [nb.bind(params).bound_impl]
    intg = LangevinIntegrator(TEMP, 1.5e-3, 0.0, masses, seed)
    steps = 10

    ctxt = co.Context(coords, v0, box, intg.impl(), [])
    with pytest.raises(RuntimeError, match="unable to find a NonbondedAllPairs potential"):
        ctxt.setup_local_md(TEMP, freeze_reference)
    ctxt = co.Context(coords, v0, box, intg.impl(), bps * 2)
    with pytest.raises(RuntimeError, match="found multiple NonbondedAllPairs potentials"):
        ctxt.setup_local_md(TEMP, freeze_reference)

    ctxt = co.Context(coords, v0, box, intg.impl(), bps)
    ctxt.setup_local_md(TEMP, freeze_reference)
    ctxt.setup_local_md(TEMP, freeze_reference)  # idempotent for equal arguments
    with pytest.raises(RuntimeError, match="local md configured with different parameters"):
        ctxt.setup_local_md(TEMP + 1, freeze_reference)
    with pytest.raises(RuntimeError, match="local md configured with different parameters"):
        ctxt.setup_local_md(TEMP, not freeze_reference)
    ref_xs, ref_boxes = ctxt.multiple_steps(steps)

    ctxt = co.Context(coords, v0, box, intg.impl(), bps)
    comp_xs, comp_boxes = ctxt.multiple_steps(steps)  # no local-MD setup at all
    np.testing.assert_array_equal(ref_xs, comp_xs)
    np.testing.assert_array_equal(ref_boxes, comp_boxes)

    ctxt = co.Context(coords, v0, box, intg.impl(), bps)
    ctxt.setup_local_md(TEMP, freeze_reference)
    ref_local_xs, ref_local_boxes = ctxt.multiple_steps_local(steps, local_idxs)
    ctxt = co.Context(coords, v0, box, intg.impl(), bps)
    if freeze_reference:  # the implicit setup is (integrator temperature, frozen reference)
        comp_local_xs, comp_local_boxes = ctxt.multiple_steps_local(steps, local_idxs)
    else:
        ctxt.setup_local_md(TEMP, freeze_reference)
        comp_local_xs, comp_local_boxes = ctxt.multiple_steps_local(steps, local_idxs)
    np.testing.assert_array_equal(ref_local_xs, comp_local_xs)
    np.testing.assert_array_equal(ref_local_boxes, comp_local_boxes)


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_local_md_with_selection_mask(co, solvated, freeze_reference):
    """multiple_steps_local_selection moves exactly the selection (and a free reference); a selection equal to what
    multiple_steps_local picked gives the same frames"""
    from timemachine_amd.lib import LangevinIntegrator

    s, coords = solvated
    N = s.num_atoms
    v0 = np.zeros_like(coords)
    local_idxs = np.arange(N - 30, N, dtype=np.int32)
    intg = LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 2023)

    a = co.Context(coords, v0, s.box, intg.impl(), make_bps(s))
    a.setup_local_md(TEMP, freeze_reference)
    xs_a, _ = a.multiple_steps_local(50, local_idxs, radius=0.8, seed=11)
    ref_atom, free = a.local_md_last_selection()
    selection = free[free != ref_atom].astype(np.int32)
    assert 30 < len(selection) < N - 1

    b = co.Context(coords, v0, s.box, intg.impl(), make_bps(s))
    b.setup_local_md(TEMP, freeze_reference)
    xs_b, _ = b.multiple_steps_local_selection(50, ref_atom, selection, radius=0.8)
    np.testing.assert_array_equal(xs_a, xs_b)
    moved = np.flatnonzero(np.any(xs_b[-1] != coords, axis=1))
    expected = np.sort(np.concatenate([selection, [ref_atom]])) if not freeze_reference else np.sort(selection)
    np.testing.assert_array_equal(moved, expected)

    # a selection of the caller's own making: everything within 0.8 nm of the last ligand atom, restrained at 1.0 nm
    # (the reference asks for selections that are plausible under the restraint, local_md_potentials.cu:149-151: atoms far
    # outside the radius would be pulled in with k (r - radius)^3)
    d = coords - coords[N - 1]
    d -= np.diagonal(s.box) * np.rint(d / np.diagonal(s.box))
    selection = np.flatnonzero((np.linalg.norm(d, axis=1) < 0.8) & (np.arange(N) != N - 1)).astype(np.int32)
    assert len(selection) > 50
    c = co.Context(coords, v0, s.box, intg.impl(), make_bps(s))
    c.setup_local_md(TEMP, freeze_reference)
    xs_c, _ = c.multiple_steps_local_selection(100, N - 1, selection, radius=1.0)
    moved = np.flatnonzero(np.any(xs_c[-1] != coords, axis=1))
    expected = np.sort(np.concatenate([selection, [N - 1]])) if not freeze_reference else selection
    np.testing.assert_array_equal(moved, expected)
    assert np.all(np.isfinite(xs_c))


@pytest.mark.parametrize("seed", [2022, 7, 123456789])
def test_selection_follows_the_oracle(co, solvated, seed):
    """the reference atom (mt19937 + uniform_int_distribution, as the reference draws it) and every atom's free / frozen
    decision are the oracle's"""
    from oracle import local_md as olm
    from timemachine_amd.lib import LangevinIntegrator

    s, coords = solvated
    N = s.num_atoms
    local_idxs = np.arange(N - 30, N, dtype=np.int32)
    radius, k = 0.6, 2000.0
    ctxt = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 5).impl(), make_bps(s))
    ctxt.multiple_steps_local(1, local_idxs, radius=radius, k=k, seed=seed)
    ref_atom, free = ctxt.local_md_last_selection()
    assert ref_atom == olm.reference_index(local_idxs, seed)
    want, margin = olm.select_free(coords, s.box, ref_atom, radius, k, TEMP, seed, freeze_reference=True)
    got = np.zeros(N, dtype=bool)
    got[free] = True
    decided = margin > 1e-5  # an exponential's last bit may differ between the device and numpy
    assert decided.sum() > N - 20
    np.testing.assert_array_equal(got[decided], want[decided])
    # the shell is probabilistic: some atoms beyond the radius are free, some are not, everybody inside is
    d = coords - coords[ref_atom]
    d -= np.diagonal(s.box) * np.rint(d / np.diagonal(s.box))
    r = np.linalg.norm(d, axis=1)
    inside = r < radius
    inside[ref_atom] = False
    assert np.all(got[inside])
    shell = (r > radius) & (r < radius + 0.25)
    assert 0 < got[shell].sum() < shell.sum()
    assert not np.any(got[r > radius + 0.6])


def test_local_md_all_pairs_on_a_subset(co, solvated):
    """a NonbondedAllPairs over a subset of the atoms (reference: free / frozen sets are intersected with it,
    local_md_potentials.cu:198-214): atoms outside the subset still move when selected, feel no nonbonded force, and the
    potential gets its subset back"""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator

    s, coords = solvated
    N = s.num_atoms
    subset = np.arange(0, N - 30, dtype=np.int32)  # the waters only
    nb = P.NonbondedAllPairs(N, s.beta, s.cutoff, atom_idxs=subset).bind(s.nb_params).to_gpu(np.float32).bound_impl
    others = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)[:-1]]
    bps = others + [nb]
    before = nb.execute(coords, s.box)
    ctxt = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.0e-3, 1.0, s.masses, 3).impl(), bps)
    xs, _ = ctxt.multiple_steps_local(20, np.array([N - 1], dtype=np.int32), radius=0.9, seed=4)
    ref_atom, free = ctxt.local_md_last_selection()
    assert ref_atom == N - 1
    assert np.any(free >= N - 30) and np.any(free < N - 30)
    np.testing.assert_array_equal(np.flatnonzero(np.any(xs[-1] != coords, axis=1)), free)
    assert np.all(np.isfinite(xs))
    after = nb.execute(coords, s.box)
    np.testing.assert_array_equal(before[0], after[0])
    assert before[1] == after[1]
    np.testing.assert_array_equal(nb.get_potential().get_atom_idxs(), subset)


@pytest.mark.parametrize("precision", [np.float32, np.float64])
def test_local_md_restraint_holds_the_free_atoms(co, solvated, precision):
    """physics: over a few thousand local steps the free atoms stay within the restraint's reach of the reference atom,
    frozen atoms never move, and alternating local / global MD keeps the system intact (finite, bonded geometry sane)"""
    from timemachine_amd.lib import LangevinIntegrator

    s, coords = solvated
    N = s.num_atoms
    local_idxs = np.arange(N - 30, N, dtype=np.int32)
    radius, k = 0.7, 10000.0
    ctxt = co.Context(coords, np.zeros_like(coords), s.box, LangevinIntegrator(TEMP, 1.5e-3, 1.0, s.masses, 9).impl(), make_bps(s, precision))
    x_prev = coords
    for it in range(4):
        xs, _ = ctxt.multiple_steps_local(500, local_idxs, radius=radius, k=k, seed=100 + it)
        ref_atom, free = ctxt.local_md_last_selection()
        x = xs[-1]
        frozen = np.setdiff1d(np.arange(N), free)
        np.testing.assert_array_equal(x[frozen], x_prev[frozen])
        d = x[free] - x[ref_atom]
        d -= np.diagonal(s.box) * np.rint(d / np.diagonal(s.box))
        r = np.linalg.norm(d, axis=1)
        # k/4 (r - radius)^4 = 10 kT at r - radius = (40 kT / k)^(1/4) = 0.32 nm: nobody gets much further than selected
        assert r.max() < radius + 0.6
        gx, _ = ctxt.multiple_steps(100)
        x_prev = gx[-1]
        assert np.all(np.isfinite(x_prev))
    # water O-H bonds are still bonds
    oh = np.linalg.norm(x_prev[1 : N - 30 : 3] - x_prev[0 : N - 30 : 3], axis=1)
    assert 0.08 < oh.min() and oh.max() < 0.115


@pytest.mark.parametrize("freeze_reference", [True, False])
def test_local_md_nonbonded_all_pairs_subset(co, freeze_reference):
    """tests/test_md.py:838-892: a Nonbonded over five random atoms of thirty (exclusions filtered with it); the reference
    atom need not be one of them.  Local MD runs, and the potential answers afterwards exactly as before."""
    from timemachine_amd.lib import LangevinIntegrator

    coords, params, pot, masses = tiny_nb_system(2022, N=30)
    rng = np.random.default_rng(2022)
    N = len(coords)
    box = np.eye(3) * 3.0
    pot.atom_idxs = np.sort(rng.choice(np.arange(N, dtype=np.int32), size=5, replace=False)).astype(np.int32)
    bps = [pot.bind(params).to_gpu(np.float32).bound_impl]
    ref_vals = [bp.execute(coords, box) for bp in bps]
    ctxt = co.Context(coords, np.zeros_like(coords), box, LangevinIntegrator(TEMP, 1.5e-3, 0.0, masses, 2022).impl(), bps)
    ctxt.setup_local_md(TEMP, freeze_reference)
    xs, boxes = ctxt.multiple_steps_local(100, np.array([N - 1], dtype=np.int32), radius=1.2)
    assert np.all(np.isfinite(xs))
    for (ref_du_dx, ref_u), bp in zip(ref_vals, bps):
        du_dx, u = bp.execute(coords, box)
        np.testing.assert_array_equal(ref_du_dx, du_dx)
        np.testing.assert_equal(ref_u, u)
        du_dx_end, _ = bp.execute(xs[-1], boxes[-1])
        assert np.all(np.isfinite(du_dx_end))
        # atoms outside the potential's subset feel nothing from it, before and after
        outside = np.setdiff1d(np.arange(N), pot.atom_idxs)
        assert not np.any(du_dx[outside]) and not np.any(du_dx_end[outside])


def test_setup_context_with_references(co, solvated):
    """tests/test_md.py:769-835: a Context keeps its integrator, bound potentials and movers alive on its own, and lets go of
    them when it goes"""
    import gc
    import weakref

    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s, coords = solvated
    N = s.num_atoms

    def build_context(barostat_interval):
        refs, bps = [], []
        for bp in ts.bound_potentials(s):
            impl = bp.to_gpu(np.float32).bound_impl
            bps.append(impl)
            refs.append(weakref.ref(impl))
        movers = []
        if barostat_interval > 0:
            groups = [list(range(3 * i, 3 * i + 3)) for i in range((N - 30) // 3)] + [list(range(N - 30, N))]
            baro = MonteCarloBarostat(N, 1.0, TEMP, groups, barostat_interval, 2022).impl(bps)
            movers.append(baro)
            refs.append(weakref.ref(baro))
        intg = LangevinIntegrator(TEMP, 1.5e-3, 0.0, s.masses, 2022).impl()
        refs.append(weakref.ref(intg))
        return co.Context(coords, np.zeros_like(coords), s.box, intg, bps, movers=movers), refs

    for interval in (0, 10):
        ctxt, refs = build_context(interval)
        gc.collect()
        if co.BINDING == "ctypes":
            assert all(r() is not None for r in refs)  # the twin's Context holds the Python objects themselves
        # (the compiled module, like the reference's, holds the C++ objects: their Python wrappers may already be gone --
        # what matters is that the Context still runs, below)
        xs, boxes = ctxt.multiple_steps(100)
        assert np.all(np.isfinite(xs)) and np.all(np.isfinite(boxes))
        assert np.all(xs[-1] != coords)
        if interval == 0:
            np.testing.assert_array_equal(boxes[-1], s.box)
        else:
            assert np.all(np.diagonal(boxes[-1]) != np.diagonal(s.box))  # the barostat changed the box
        del ctxt
        gc.collect()
        assert all(r() is None for r in refs)
