"""GPU tests (``-m gpu``) of MonteCarloBarostat after the reference's tests/test_barostat.py: constructor validation (:21-70),
zero interval (:137-176), a barostat over a subset of the molecules (:179-241), determinism for a seed (:244-302), pressure
dependence of the volume (:305-357).  Synthetic solvated-ligand systems (timemachine_amd.testsystems)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


@pytest.fixture(scope="module")
def relaxed():
    """~770 waters + a 20-atom ligand, relaxed at constant volume; groups = molecules"""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops

    custom_ops.set_device(0)
    s = ts.small_solvated_ligand()
    N = s.num_atoms
    groups = [list(range(3 * k, 3 * k + 3)) for k in range((N - 20) // 3)] + [list(range(N - 20, N))]
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    nvt = custom_ops.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), bps)
    nvt.multiple_steps(1500, 0)
    return s, groups, nvt.get_x_t(), nvt.get_v_t()


def make_bps(s):
    from timemachine_amd import testsystems as ts

    return [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]


def volume(box):
    return float(np.prod(np.diagonal(box)))


def test_barostat_validation(co, relaxed):
    s, groups, x, v = relaxed
    N, u_impls = s.num_atoms, make_bps(s)
    with pytest.raises(RuntimeError, match="interval must be greater than 0"):
        co.MonteCarloBarostat(N, 1.0, 300.0, [[0, 1]], -1, u_impls, 2023, True, 0.0)
    with pytest.raises(RuntimeError, match="Grouped indices must be between 0 and N"):
        co.MonteCarloBarostat(N, 1.0, 300.0, [[0, N + 1]], 3, u_impls, 2023, True, 0.0)
    with pytest.raises(RuntimeError, match="Grouped indices must be between 0 and N"):
        co.MonteCarloBarostat(N, 1.0, 300.0, [[-1, 0]], 3, u_impls, 2023, True, 0.0)
    with pytest.raises(RuntimeError, match="All grouped indices must be unique"):
        co.MonteCarloBarostat(N, 1.0, 300.0, [[0, 1], [1, 2]], 3, u_impls, 2023, True, 0.0)


def test_barostat_zero_interval(co, relaxed):
    s, groups, x, v = relaxed
    u_impls = make_bps(s)
    with pytest.raises(RuntimeError):
        co.MonteCarloBarostat(s.num_atoms, 1.0, 300.0, groups, 0, u_impls, 2021, True, 0.0)
    baro = co.MonteCarloBarostat(s.num_atoms, 1.0, 300.0, groups, 1, u_impls, 2021, True, 0.0)
    with pytest.raises(RuntimeError):
        baro.set_interval(0)


def test_barostat_partial_group_idxs(co, relaxed):
    """a barostat that only knows half of the molecules runs (the rest keep their coordinates through volume moves)"""
    from timemachine_amd.lib import LangevinIntegrator

    s, groups, x, v = relaxed
    u_impls = make_bps(s)
    baro = co.MonteCarloBarostat(s.num_atoms, 1.0, 300.0, groups[len(groups) // 2:], 3, u_impls, 2021, True, 0.0)
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 2021).impl(), u_impls, movers=[baro])
    ctxt.multiple_steps(3 * 100)
    assert np.all(np.isfinite(ctxt.get_x_t())) and np.all(np.isfinite(ctxt.get_box()))


@pytest.mark.parametrize("iterations", [20, 300])
def test_barostat_is_deterministic(co, relaxed, iterations):
    """the same seeds give the same box after the same number of steps, bit for bit -- and a box that moved"""
    from timemachine_amd.lib import LangevinIntegrator

    s, groups, x, v = relaxed
    boxes = []
    for _ in range(2):
        u_impls = make_bps(s)
        baro = co.MonteCarloBarostat(s.num_atoms, 1.0, 300.0, groups, 3, u_impls, 2021, True, 0.0)
        ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 2021).impl(), u_impls, movers=[baro])
        ctxt.multiple_steps(iterations * 3)
        boxes.append(ctxt.get_box())
    assert volume(boxes[0]) != volume(s.box)
    np.testing.assert_array_equal(boxes[0], boxes[1])


def test_barostat_varying_pressure(co, relaxed):
    """tests/test_barostat.py:305-357: at 1000 bar the box ends smaller than at 1 bar from the same start"""
    from timemachine_amd.lib import LangevinIntegrator

    s, groups, x, v = relaxed
    vols = {}
    for pressure in (1.0, 1000.0):
        u_impls = make_bps(s)
        baro = co.MonteCarloBarostat(s.num_atoms, pressure, 300.0, groups, 3, u_impls, 2019, True, 0.0)
        ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 2019).impl(), u_impls, movers=[baro])
        ctxt.multiple_steps(3000)
        vols[pressure] = volume(ctxt.get_box())
    assert vols[1000.0] < vols[1.0], vols


def test_barostat_recentering_upon_acceptance(co, relaxed):
    """tests/test_barostat.py:360-423: a standalone move() either leaves coordinates and box untouched (rejected / off-interval)
    or returns a scaled box in which every molecule's centroid lies inside the home cell and molecules are whole."""
    from timemachine_amd.lib import LangevinIntegrator

    s, groups, x, v = relaxed
    u_impls = make_bps(s)
    baro = co.MonteCarloBarostat(s.num_atoms, 1.0, 300.0, groups, 10, u_impls, 2023, True, 0.0)
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 2023).impl(), u_impls, movers=[baro])
    ctxt.multiple_steps(1000)
    num_accepted = 0
    for _ in range(100):
        ctxt.multiple_steps(100)
        x_t, box_t = ctxt.get_x_t(), ctxt.get_box()
        new_x_t, new_box_t = baro.move(x_t, box_t)
        if not np.all(box_t == new_box_t):
            L = np.diagonal(new_box_t)
            for atom_idxs in groups:
                mol = new_x_t[atom_idxs]
                # whole molecule: every atom within half a box of the first one without re-imaging
                assert np.all(np.abs(mol - mol[0]) < 0.5 * L)
                c = mol.mean(axis=0)
                assert np.all(c > -1e-6) and np.all(c < L + 1e-6)
            num_accepted += 1
        else:
            np.testing.assert_array_equal(new_x_t, x_t)
            np.testing.assert_array_equal(new_box_t, box_t)
    assert num_accepted > 0


def test_molecular_ideal_gas(co, relaxed):
    """tests/test_barostat.py:426-527 (after OpenMM's testIdealGas): with the nonbonded terms removed the molecules are an ideal
    gas of rigid-ish bodies, and the barostat's acceptance rule must give <V> = N_mol kT / P -- to 1 % at 300, 600 and 1000 K
    and 100 bar.  This pins the Metropolis criterion (N_mol ln(V'/V) term, P dV term, molecule-centroid scaling) physically."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.constants import AVOGADRO, BOLTZ
    from timemachine_amd.lib import LangevinIntegrator

    BAR_TO_KJ_PER_NM3 = 1e-25
    s, groups, x_relaxed, _ = relaxed
    bonded = [bp for bp in ts.bound_potentials(s) if not isinstance(bp.potential, (P.Nonbonded, P.NonbondedInteractionGroup))]
    u_impls = [bp.to_gpu(np.float32).bound_impl for bp in bonded]
    n_mols = len(groups)
    pressure, interval, n_steps = 100.0, 5, 10000
    temperatures = np.array([300.0, 600.0, 1000.0])
    expected = n_mols * BOLTZ * temperatures / (pressure * AVOGADRO * BAR_TO_KJ_PER_NM3)
    rng = np.random.default_rng(2021)
    actual = []
    for T, v_expected in zip(temperatures, expected):
        # start 2 % off the expected volume: molecule centroids scaled with the box about its centre
        scale = (1.02 * v_expected / np.prod(np.diagonal(s.box))) ** (1.0 / 3.0)
        box = s.box * scale
        x = x_relaxed.copy()
        center = 0.5 * np.diagonal(s.box)
        for g in groups:
            c = x[g].mean(axis=0)
            x[g] += (c - center) * scale + center * scale - c
        v0 = rng.normal(size=x.shape) * np.sqrt(BOLTZ * T / s.masses)[:, None]
        baro = co.MonteCarloBarostat(s.num_atoms, pressure, T, groups, interval, u_impls, 2021, True, 0.0)
        ctxt = co.Context(x, v0, box, LangevinIntegrator(T, 1.5e-3, 1.0, s.masses, 2021).impl(), u_impls, movers=[baro])
        vols = []
        for _ in range(n_steps // interval):
            ctxt.multiple_steps(interval)
            vols.append(np.prod(np.diagonal(ctxt.get_box())))
        actual.append(np.mean(vols[len(vols) // 2:]))
    np.testing.assert_allclose(actual, expected, rtol=1e-2)


def test_barostat_scaling_behavior(co, relaxed):
    """tests/test_barostat.py:579-665: the volume scale factor can be read and set; adaptation shrinks an absurd factor,
    leaves a zero factor at zero when switched off, and moves it again when switched back on; the constructor's initial
    factor and flag are kept."""
    from timemachine_amd.lib import LangevinIntegrator

    s, groups, x, v = relaxed
    u_impls = make_bps(s)
    baro = co.MonteCarloBarostat(s.num_atoms, 1.013, 300.0, groups, 3, u_impls, 2021, True, 0.0)
    assert baro.get_volume_scale_factor() == 0.0
    assert baro.get_adaptive_scaling()
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 2021).impl(), u_impls, movers=[baro])
    ctxt.multiple_steps(15)
    scaling = baro.get_volume_scale_factor()
    assert scaling > 0
    bad = 0.5 * volume(s.box)
    baro.set_volume_scale_factor(bad)
    assert baro.get_volume_scale_factor() == bad
    ctxt.multiple_steps(100)
    assert bad > baro.get_volume_scale_factor()
    baro.set_volume_scale_factor(scaling)
    assert scaling == baro.get_volume_scale_factor()
    baro.set_volume_scale_factor(0.0)
    baro.set_adaptive_scaling(False)
    assert not baro.get_adaptive_scaling()
    ctxt.multiple_steps(100)
    assert baro.get_volume_scale_factor() == 0.0
    baro.set_adaptive_scaling(True)
    assert baro.get_adaptive_scaling()
    ctxt.multiple_steps(100)
    assert baro.get_volume_scale_factor() != 0.0
    baro = co.MonteCarloBarostat(s.num_atoms, 1.013, 300.0, groups, 3, u_impls, 2021, False, 1.23)
    assert not baro.get_adaptive_scaling()
    assert baro.get_volume_scale_factor() == 1.23


@pytest.mark.parametrize("precision", [np.float32, np.float64])
@pytest.mark.parametrize("interval,pressure,padding", [(1, 1.0, 0.18), (3, 400.0, 0.1), (25, 1.0, 0.18)])
def test_box_scaling_keeps_the_neighbor_list_valid(co, relaxed, precision, interval, pressure, padding):
    """Potentials a barostat works on follow small box changes without rebuilding their neighbor list: the list stays a
    superset of the pairs inside the cutoff, so trajectories, boxes and the Metropolis decisions are bit-identical to a run in
    which every box change rebuilds -- attempt after attempt, at high pressure (steady compression) as well, with fewer builds."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s, groups, x, v = relaxed
    N = s.num_atoms

    def run(reuse):
        co.debug_set_box_scaling_reuse(reuse)
        static_before = co.debug_set_static_list_max_k(0)  # this test is about list REBUILDS: a static, complete list has none
        try:
            bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision, nblist_padding=padding)]
            baro = MonteCarloBarostat(N, pressure, 300.0, groups, interval, 5).impl(bps)
            ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.0e-3, 1.0, s.masses, 9).impl(), bps, movers=[baro])
            xs, boxes = ctxt.multiple_steps(600, 20)
            nb = bps[-1].get_potential().get_potentials()[0]
            return xs, boxes, baro.get_counters(), nb.get_build_count()
        finally:
            co.debug_set_box_scaling_reuse(True)
            co.debug_set_static_list_max_k(static_before)

    xs_a, boxes_a, counters_a, builds_a = run(True)
    xs_b, boxes_b, counters_b, builds_b = run(False)
    np.testing.assert_array_equal(boxes_a, boxes_b)
    np.testing.assert_array_equal(xs_a, xs_b)
    assert counters_a == counters_b and counters_a[1] == 600 // interval and counters_a[0] > 0
    assert not np.array_equal(boxes_a[-1], s.box)
    assert builds_a < builds_b, (builds_a, builds_b)


@pytest.mark.parametrize("precision", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["listed_2k", "static_2k", "dhfr_shaped", "big_molecule"])
def test_attempts_on_the_current_list_are_bitwise_the_reference_shaped_attempts(co, relaxed, which, precision):
    """Inside a Context a barostat attempt runs on the nonbonded potential's CURRENT neighbor list and sorted records whenever the
    last MD step left them in place (csrc/barostat.hip, "the fast path": proposal + list-validity test in one launch, two energy
    launches, decision + commit of an accepted proposal into the pre-gathered state).  Energies are the same integers, so every
    decision, every box and every trajectory must equal the reference-shaped attempt's (barostat.cu:154-246: copy, centroids,
    rescale, two full evaluations, decision) bit for bit -- on the listed pipeline (rebuilds, re-sorts, proposals that the list must
    be rebuilt for), on the static complete list of small systems, at the bench workload's size, at high pressure (steady
    compression: the accumulated box scale runs into its limit and forces rebuilds), and with one molecule larger than the
    in-thread centroid limit (its centroid comes from the segmented-scan kernel)."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    if which == "dhfr_shaped":
        s = ts.dhfr_shaped_box()
        x, v = s.coords.astype(np.float32).astype(np.float64), np.zeros_like(s.coords)
        groups, n_steps, interval, dt, friction, pressure, padding, static_k = ts.molecule_groups(s), 150, 5, 1.0e-3, 10.0, 1.0, 0.18, 0
    else:
        s, groups, x, v = relaxed
        n_steps, interval, dt, friction, padding = 400, 3, 2.0e-3, 1.0, 0.1
        pressure = 400.0 if which == "listed_2k" else 1.0
        static_k = 4608 if which == "static_2k" else 0
        if which == "big_molecule":  # the first 150 waters as ONE rigid group of 450 atoms (> the in-thread centroid limit)
            groups = [list(range(450))] + [g for g in groups if g[0] >= 450]
    N = s.num_atoms

    def run(fast):
        fast_before = co.debug_set_barostat_fast_path(fast)
        static_before = co.debug_set_static_list_max_k(static_k)
        try:
            bps = [bp.to_gpu(precision).bound_impl for bp in ts.bound_potentials(s, precision, nblist_padding=padding)]
            baro = MonteCarloBarostat(N, pressure, 300.0, groups, interval, 11).impl(bps)
            ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 9).impl(), bps, movers=[baro])
            xs, boxes = ctxt.multiple_steps(n_steps, interval)
            ctxt.multiple_steps(7, 0)  # (a call boundary: the first step of a call gathers for itself; attempts inside take either path)
            return xs, boxes, ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box(), baro.get_counters(), baro.get_volume_scale_factor(), baro.get_attempt_paths()
        finally:
            co.debug_set_barostat_fast_path(fast_before)
            co.debug_set_static_list_max_k(static_before)

    a, b = run(True), run(False)
    for u, w in zip(a[:5], b[:5]):
        np.testing.assert_array_equal(u, w)
    assert a[5] == b[5] and a[6] == b[6]
    attempts, fast = a[7]
    assert attempts == (n_steps + 7) // interval and fast >= 0.9 * attempts, a[7]  # all but the attempts that meet a Hilbert re-sort
    assert b[7] == (attempts, 0)
    assert not np.array_equal(a[4], s.box)  # moves were accepted


def test_a_proposal_beyond_the_lists_reach_is_reported_not_decided_in_silence(co):
    """The fast path evaluates a proposal on the nonbonded potential's CURRENT list (rebuilt from the current geometry if the proposal
    fails the list's validity test).  A proposal that scales the box by several per cent can hold pairs inside the cutoff that even
    that fresh list does not (it reaches cutoff + padding in the current geometry): `k_barostat_propose_probe` notes the first such
    attempt in host-visible memory and the stepping call ends with an error instead of a trajectory decided on wrong energies.  With
    the reference-shaped attempts (two full evaluations, each listing its own geometry) the same barostat runs; and moves of ordinary
    size never trip the check (every other fast-path test of this suite steps through `after_wait`)."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.config4_solvated_ligand()  # 6 318 atoms: the listed pipeline (a static complete list holds every pair and has no reach to exceed)
    N = s.num_atoms
    groups = ts.molecule_groups(s)

    def make(volume_scale):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
        baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 5, 3, adaptive_scaling_enabled=False, initial_volume_scale_factor=volume_scale).impl(bps)
        ctxt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 2).impl(), bps, movers=[baro])
        return ctxt, baro

    volume = float(np.prod(np.diag(s.box)))
    ctxt, baro = make(0.6 * volume)  # |s - 1| up to 17 %: (1.2 + ...) / 0.83 is far beyond cutoff + padding = 1.3
    with pytest.raises(RuntimeError, match="scaled further than the nonbonded potential's neighbor list reaches"):
        for _ in range(20):
            ctxt.multiple_steps(5, 0)
    before = co.debug_set_barostat_fast_path(False)
    try:
        ctxt, baro = make(0.6 * volume)
        ctxt.multiple_steps(40, 0)
        assert baro.get_attempt_paths() == (8, 0) and np.all(np.isfinite(ctxt.get_x_t()))
    finally:
        co.debug_set_barostat_fast_path(before)
    ctxt, baro = make(0.01 * volume)  # an ordinary move size: |s - 1| <= 0.33 %
    ctxt.multiple_steps(100, 0)
    assert baro.get_attempt_paths() == (20, 20)
