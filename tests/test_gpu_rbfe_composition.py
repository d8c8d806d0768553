"""GPU tests (``-m gpu``) of the composition the reference's RBFE / AHFE windows actually have -- ``HostGuestSystem``
(timemachine/fe/system.py:133-146, built at fe/single_topology.py:2100-2154): bond, angle, proper, improper, chiral_atom,
ligand-ligand ``NonbondedPairListPrecomputed``, host-host ``Nonbonded(atom_idxs=host)`` and the ligand-environment
``NonbondedInteractionGroup`` -- two producers of tile work with two parameter arrays.

From round 6 on a forces-only or energy-only plan runs the two as ONE pipeline (csrc/engine.hpp: the all-pairs potential's merged
carrier; ``ForcePlan::merge_producers``): one list, one tile launch, one sorted hand-over to the integrator, the barostat's attempts
on that list.  What is proved here: the merged pipeline gives the bits of the separate potentials (forces, energies, trajectories,
NPT trajectories), on the listed pipeline and on static lists, in both precisions, alone and grouped; its forces agree with the
oracle at config-5 size; and WHICH path ran is asserted (merged evaluations, fast barostat attempts), not assumed.
Reference semantics: tests/nonbonded/test_consistency.py (a HostGuestSystem's parts add up), tests/test_barostat.py."""
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


def _system(which):
    from timemachine_amd import testsystems as ts

    if which == "config2":  # 2.2k atoms, 20-atom ligand (20 guest rows + 12 holes in the merged order)
        return ts.small_solvated_ligand(lamb=0.3), 20
    if which == "config4":  # 6.3k atoms, 30-atom ligand
        return ts.config4_solvated_ligand(lamb=0.3), 30
    if which == "config5":  # 31k atoms, 40-atom ligand (two guest row blocks)
        return ts.config5_complex_sized(lamb=0.3), 40
    raise AssertionError(which)


def _all_pairs_of(impl):
    name = type(impl).__name__
    if name.startswith("NonbondedAllPairs"):
        return impl
    if hasattr(impl, "get_potentials"):
        for c in impl.get_potentials():
            r = _all_pairs_of(c)
            if r is not None:
                return r
    return None


def _host_all_pairs(bound_impls):
    for bp in bound_impls:
        r = _all_pairs_of(bp.get_potential())
        if r is not None:
            return r
    raise AssertionError("no NonbondedAllPairs among the bound potentials")


class _Switches:
    """process-wide A/B switches set for a block and restored after it"""

    def __init__(self, co, merge=True, static_k=None, fast=None):
        self.co, self.merge, self.static_k, self.fast = co, merge, static_k, fast

    def __enter__(self):
        self.m0 = self.co.debug_set_merge_producers(self.merge)
        self.s0 = self.co.debug_set_static_list_max_k(self.static_k) if self.static_k is not None else None
        self.f0 = self.co.debug_set_barostat_fast_path(self.fast) if self.fast is not None else None

    def __exit__(self, *exc):
        self.co.debug_set_merge_producers(self.m0)
        if self.s0 is not None:
            self.co.debug_set_static_list_max_k(self.s0)
        if self.f0 is not None:
            self.co.debug_set_barostat_fast_path(self.f0)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k,env_scale", [("config2", 4608, None), ("config2", 0, 0.9), ("config4", 0, None), ("config5", 0, 0.9)])
def test_merged_evaluation_is_bitwise_the_sum_of_the_parts(co, P, which, static_k, env_scale, precision):
    """One SummedPotential of the whole state: forces-only and energy-only calls (the forms MD and the barostat / HREX use) with
    the two tile producers merged, against the same calls with merging switched off, against the parts evaluated one by one.
    ``env_scale``: the interaction group sees the host's charges rescaled (an environment BCC handle, single_topology.py:2040):
    the host atoms' two records then differ."""
    from timemachine_amd import testsystems as ts

    s, n_lig = _system(which)
    state = ts.rbfe_shaped_state(s, n_lig, env_charge_scale=env_scale)
    rng = np.random.default_rng(5)
    x = s.coords + rng.normal(0.0, 0.003, s.coords.shape)
    if precision == np.float32:
        x = x.astype(np.float32).astype(np.float64)
    box = s.box
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])

    def summed():
        return P.SummedPotential([p for p, _ in state], [q for _, q in state], parallel=False).to_gpu(precision).unbound_impl

    with _Switches(co, merge=True, static_k=static_k):
        impl = summed()
        f_merged = impl.execute_raw(x, flat, box, True, False, False)[0]
        u_merged = impl.execute_raw(x, flat, box, False, False, True)[2]
        f_again = impl.execute_raw(x + 0.0, flat, box, True, False, False)[0]  # a second call: the list exists, nothing is forced
        calls, tiles, builds = _all_pairs_of(impl).get_merged_stats()
        assert calls == 3 and tiles > 0 and builds >= 1, (calls, tiles, builds)
        # a full call (du/dp wanted) is not the carrier's business: the parts run, and give the forces-only bits
        full = impl.execute_raw(x, flat, box, True, True, True)
        assert _all_pairs_of(impl).get_merged_stats()[0] == 3
    with _Switches(co, merge=False, static_k=static_k):
        impl = summed()
        f_apart = impl.execute_raw(x, flat, box, True, False, False)[0]
        u_apart = impl.execute_raw(x, flat, box, False, False, True)[2]
        assert _all_pairs_of(impl).get_merged_stats()[0] == 0
        acc = np.zeros_like(f_apart)
        u_sum = 0
        for pot, prm in state:
            r = pot.to_gpu(precision).unbound_impl.execute_raw(x, np.asarray(prm, dtype=np.float64), box, True, False, True)
            acc += r[0]
            u_sum += int(r[2])
    np.testing.assert_array_equal(f_merged, f_apart)
    np.testing.assert_array_equal(f_again, f_apart)
    np.testing.assert_array_equal(acc, f_apart)
    np.testing.assert_array_equal(full[0], f_apart)
    assert int(u_merged) == int(u_apart) == u_sum == int(full[2])
    assert np.any(f_merged[-n_lig:] != 0) and np.any(f_merged[:16] != 0)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_config5_sized_composition_against_the_oracle(co, P, precision):
    """31 540 atoms on the listed pipeline, merged: the ligand atoms' and 160 sampled host atoms' nonbonded forces (host-host all
    pairs minus exclusions + ligand-host group, each with its own parameter array, charges of the group's host side rescaled)
    against the oracle's pair function with the reference's semantics (potentials/nonbonded.py:221-399, 403-481), 1e-8 / 1e-4 of the
    force norm; the bonded / chiral / precomputed terms against the oracle over the whole system."""
    import torch

    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts

    s, n_lig = _system("config5")
    N, n_host = s.num_atoms, s.num_atoms - n_lig
    state = ts.rbfe_shaped_state(s, n_lig, env_charge_scale=0.9)
    rng = np.random.default_rng(23)
    x = s.coords + rng.normal(0.0, 0.004, s.coords.shape)
    if precision == np.float32:
        x = x.astype(np.float32).astype(np.float64)
    box = s.box
    f64 = precision == np.float64
    tol = 1e-8 if f64 else 1e-4
    nb_host, host_params = state[6]
    nb_group, ixn_params = state[7]
    with _Switches(co, merge=True, static_k=0):
        pots = [state[6], state[7]]
        impl = P.SummedPotential([p for p, _ in pots], [q for _, q in pots], parallel=False).to_gpu(precision).unbound_impl
        flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in pots])
        raw = impl.execute_raw(x, flat, box, True, False, False)[0]
        assert _all_pairs_of(impl).get_merged_stats()[0] == 1
    du_dx = raw.view(np.int64).astype(np.float64) / 2.0**36
    with np.errstate(over="ignore"):
        assert np.all(raw.sum(axis=0, dtype=np.uint64) == 0)  # Newton's third law, exactly

    sample = np.sort(np.concatenate([rng.choice(n_host, 160, replace=False), np.arange(n_host, N)]))
    row_of = {int(i): r for r, i in enumerate(sample)}
    keep_q, keep_lj = np.ones((len(sample), N)), np.ones((len(sample), N))
    for (i, j), (sq, slj) in zip(nb_host.exclusion_idxs, nb_host.scale_factors):
        for a_, b_ in ((int(i), int(j)), (int(j), int(i))):
            if a_ in row_of:
                keep_q[row_of[a_], b_] = 1.0 - sq
                keep_lj[row_of[a_], b_] = 1.0 - slj
    p1, p2 = torch.tensor(host_params), torch.tensor(ixn_params)
    bt = torch.tensor(np.diagonal(box).copy())
    xt = torch.tensor(x)
    others = torch.arange(N)
    is_host = others < n_host
    ref = np.zeros((len(sample), 3))
    for k0 in range(0, len(sample), 40):
        idx = sample[k0 : k0 + 40]
        ti = torch.tensor(idx)
        xi = torch.tensor(x[idx], requires_grad=True)
        d3 = rp.delta_r(xi[:, None, :], xt[None, :, :], bt)
        d2 = (d3 * d3).sum(-1)
        d2 = torch.where(ti[:, None] != others[None, :], d2, torch.full_like(d2, 1e6))
        dij = torch.sqrt(d2)
        row_host = (ti < n_host)[:, None]
        # host row x host column: the all-pairs potential's parameters; exactly one side the ligand: the group's; ligand x ligand: neither
        hh = row_host & is_host[None, :]
        hl = row_host ^ is_host[None, :]
        total = 0.0
        for mask, pp, kq, kl in ((hh, p1, torch.tensor(keep_q[k0 : k0 + 40]), torch.tensor(keep_lj[k0 : k0 + 40])), (hl, p2, 1.0, 1.0)):
            # 4-D distance: w differs between ligand and host in the group's parameters (lambda = 0.3); zero everywhere in the host's
            dw = pp[idx, 3][:, None] - pp[None, :, 3]
            d4 = torch.sqrt(d2 + dw * dw)
            lj, es = rp._pair_energies(d4, pp[idx, 0][:, None] * pp[None, :, 0], pp[idx, 1][:, None] + pp[None, :, 1], pp[idx, 2][:, None] * pp[None, :, 2], s.beta, s.cutoff)
            m = mask.to(lj.dtype)
            total = total + (lj * kl * m).sum() + (es * kq * m).sum()
        ref[k0 : k0 + 40] = torch.autograd.grad(total, xi)[0].numpy()
    norms = np.maximum(np.linalg.norm(ref, axis=1, keepdims=True), 1.0)
    assert (np.abs(ref - du_dx[sample]) / norms).max() <= tol
    assert np.linalg.norm(ref[-n_lig:], axis=1).max() > 10.0  # the ligand is pulled by its environment

    # the rest of the state, over the whole system
    for k, ref_fn, extra in ((0, rp.harmonic_bond, ()), (1, rp.harmonic_angle, ()), (2, rp.periodic_torsion, ()), (3, rp.periodic_torsion, ()), (4, rp.chiral_atom_restraint, ())):
        pot, prm = state[k]
        g = pot.to_gpu(precision).unbound_impl.execute(x, np.asarray(prm, dtype=np.float64), box)
        ref_u, ref_dx, _ = ref_fn(x, np.asarray(prm, dtype=np.float64), box, pot.idxs)
        brt = 1e-7 if f64 else 2e-4
        assert abs(g[2] - ref_u) <= brt * max(1.0, abs(ref_u)), type(pot).__name__
        nrm = np.maximum(np.linalg.norm(ref_dx, axis=1, keepdims=True), 100.0 if not f64 else 1.0)
        assert (np.abs(ref_dx - g[0]) / nrm).max() <= brt, type(pot).__name__
    pot, prm = state[5]
    g = pot.to_gpu(precision).unbound_impl.execute(x, prm, box)
    ref_u, ref_dx, _ = rp.nonbonded_pair_list_precomputed(x, prm, box, pot.idxs, s.beta, s.cutoff)
    assert abs(g[2] - ref_u) <= (1e-8 if f64 else 1e-4) * max(1.0, abs(ref_u))
    nrm = np.maximum(np.linalg.norm(ref_dx, axis=1, keepdims=True), 1.0)
    assert (np.abs(ref_dx - g[0]) / nrm).max() <= tol


def _context(co, s, n_lig, precision, padding, v0, seed=7, dt=1.5e-3, env_scale=None, barostat=None):
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    bps = [bp.to_gpu(precision).bound_impl for bp in ts.rbfe_bound_potentials(s, n_lig, nblist_padding=padding, env_charge_scale=env_scale)]
    movers = []
    baro = None
    if barostat is not None:
        interval, pressure, bseed = barostat
        baro = MonteCarloBarostat(s.num_atoms, pressure, 300.0, ts.molecule_groups(s), interval, bseed).impl(bps)
        movers = [baro]
    ctxt = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, dt, 1.0, s.masses, seed).impl(), bps, movers=movers)
    return ctxt, bps, baro


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k,n_steps", [("config2", 4608, 230), ("config2", 0, 230), ("config4", 0, 230), ("config5", 0, 120)])
def test_md_with_merged_producers_is_bitwise_md_with_separate_ones(co, P, which, static_k, n_steps, precision):
    """The reference-shaped window stepped with the merged carrier (one deferred producer: the slot-ordered update kernel, no
    gather launch, block bounds from the update kernel) against the same window with merging off (two producers: atom-order update,
    two hand-overs) and against a context whose coordinates are re-set before every step (the full gather path of the carrier).
    Across list rebuilds and Hilbert re-sorts of the merged order (every 100 steps), parameters swapped and restored in between."""
    s, n_lig = _system(which)
    rng = np.random.default_rng(3)
    v0 = rng.normal(size=s.coords.shape) * 0.2
    every = 10

    def run(merge, mode):
        with _Switches(co, merge=merge, static_k=static_k):
            ctxt, bps, _ = _context(co, s, n_lig, precision, 0.1, v0, env_scale=0.9)
            frames = []
            if mode == "one_call":
                xs, _ = ctxt.multiple_steps(n_steps, every)
                frames = list(xs)
            else:
                ixn = bps[-1]
                prm = None
                for k in range(n_steps):
                    if mode == "reset":
                        ctxt.set_x_t(ctxt.get_x_t())
                    elif k % 13 == 5:  # the group's parameters replaced behind the same device pointer, then restored
                        from timemachine_amd import testsystems as ts

                        prm = np.asarray(ts.rbfe_shaped_state(s, n_lig, env_charge_scale=0.9)[7][1], dtype=np.float64)
                        ixn.set_params((prm * 0.5).reshape(-1))
                        ixn.set_params(prm.reshape(-1))
                    ctxt.step()
                    if (k + 1) % every == 0:
                        frames.append(ctxt.get_x_t())
            stats = _host_all_pairs(bps).get_merged_stats()
            return np.array(frames), ctxt.get_v_t(), stats

    ref, v_ref, stats_ref = run(False, "one_call")
    assert np.all(np.isfinite(ref)) and stats_ref[0] == 0
    merged, v_m, stats_m = run(True, "one_call")
    assert stats_m[0] == n_steps and stats_m[2] >= (1 if static_k else 3), stats_m  # every step on the carrier; its list was rebuilt
    np.testing.assert_array_equal(merged, ref)
    np.testing.assert_array_equal(v_m, v_ref)
    if which != "config5":
        for mode in ("reset", "swap"):
            xs, v, stats = run(True, mode)
            np.testing.assert_array_equal(xs, ref)
            np.testing.assert_array_equal(v, v_ref)
            assert stats[0] == n_steps


def test_a_parameter_change_of_the_group_shows_in_merged_md(co, P):
    """the carrier's pre-gathered records hold the group's parameters too: a set_params on the interaction group that is NOT undone
    must change the trajectory (stale second records would hide it)"""
    from timemachine_amd import testsystems as ts

    s, n_lig = _system("config2")
    v0 = np.zeros_like(s.coords)
    with _Switches(co, merge=True, static_k=0):
        a, bps_a, _ = _context(co, s, n_lig, np.float32, 0.1, v0)
        a.multiple_steps(20, 0)
        prm = np.asarray(ts.rbfe_shaped_state(s, n_lig)[7][1], dtype=np.float64)
        bps_a[-1].set_params((prm * np.array([0.0, 1.0, 1.0, 1.0])).reshape(-1))  # the group's charges off
        a.multiple_steps(20, 0)
        b, bps_b, _ = _context(co, s, n_lig, np.float32, 0.1, v0)
        b.multiple_steps(40, 0)
        assert _host_all_pairs(bps_a).get_merged_stats()[0] == 40
    assert not np.array_equal(a.get_x_t(), b.get_x_t())


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k", [("config2", 0), ("config4", 0)])
def test_grouped_stepping_of_merged_windows_is_bitwise_stepping_alone(co, P, which, static_k, precision):
    """three reference-shaped windows (different lambda: different group parameters) stepped together by multiple_steps_group, a
    barostat in each, against each stepped alone"""
    from timemachine_amd import testsystems as ts

    n_steps = 130
    lambdas = [0.0, 0.3, 0.7]

    def make(lamb):
        s = ts.small_solvated_ligand(lamb=lamb) if which == "config2" else ts.config4_solvated_ligand(lamb=lamb)
        n_lig = 20 if which == "config2" else 30
        rng = np.random.default_rng(11)
        v0 = rng.normal(size=s.coords.shape) * 0.2
        return _context(co, s, n_lig, precision, 0.1, v0, barostat=(10, 1.0, 5))

    with _Switches(co, merge=True, static_k=static_k):
        alone = []
        for lamb in lambdas:
            ctxt, bps, baro = make(lamb)
            ctxt.multiple_steps(n_steps, 0)
            alone.append((ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box(), baro.get_attempt_paths()))
        grouped = [make(lamb) for lamb in lambdas]
        co.multiple_steps_group([g[0] for g in grouped], n_steps)
        for (ctxt, bps, baro), ref in zip(grouped, alone):
            np.testing.assert_array_equal(ctxt.get_x_t(), ref[0])
            np.testing.assert_array_equal(ctxt.get_v_t(), ref[1])
            np.testing.assert_array_equal(ctxt.get_box(), ref[2])
            assert baro.get_attempt_paths() == ref[3]
            assert _host_all_pairs(bps).get_merged_stats()[0] >= n_steps
            attempts, fast = baro.get_attempt_paths()
            assert attempts == n_steps // 10 and fast >= attempts - 2, (attempts, fast)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k,pressure", [("config2", 4608, 1.0), ("config2", 0, 300.0), ("config4", 0, 1.0), ("config5", 0, 1.0)])
def test_npt_on_the_merged_carrier_fast_path_is_bitwise_the_reference_shaped_path(co, P, which, static_k, pressure, precision):
    """A MonteCarloBarostat in a reference-shaped window.  Three runs that must agree bit for bit in coordinates, velocities, boxes,
    acceptance counters and volume scale: (a) merged carrier + attempts on its current list (the fast path: its DUAL launch evaluates
    host-host AND ligand-host pairs of both geometries); (b) merged carrier + reference-shaped attempts (barostat.cu:154-246: two full
    evaluations); (c) no merging, reference-shaped attempts (rounds 1-5).  And WHICH path ran is asserted: in (a) every attempt but
    those that meet a Hilbert re-sort took the fast path; in (b), (c) none; without merging the fast path never applies to this
    composition (two tile producers)."""
    s, n_lig = _system(which)
    n_steps, interval = (90, 5) if which == "config5" else (300, 4)
    rng = np.random.default_rng(17)
    v0 = rng.normal(size=s.coords.shape) * 0.2

    def run(merge, fast):
        with _Switches(co, merge=merge, static_k=static_k, fast=fast):
            ctxt, bps, baro = _context(co, s, n_lig, precision, 0.18 if which == "config5" else 0.1, v0, env_scale=0.9, barostat=(interval, pressure, 13))
            xs, boxes = ctxt.multiple_steps(n_steps, interval)
            ctxt.multiple_steps(5, 0)
            return xs, boxes, ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box(), baro.get_counters(), baro.get_volume_scale_factor(), baro.get_attempt_paths(), _host_all_pairs(bps).get_merged_stats()

    a, b, c = run(True, True), run(True, False), run(False, True)
    for other in (b, c):
        for u, w in zip(a[:5], other[:5]):
            np.testing.assert_array_equal(u, w)
        assert a[5] == other[5] and a[6] == other[6]
    attempts, fast = a[7]
    assert attempts == (n_steps + 5) // interval and fast >= 0.9 * attempts, a[7]
    assert b[7] == (attempts, 0) and c[7] == (attempts, 0), (b[7], c[7])
    assert a[8][0] >= n_steps and c[8][0] == 0
    assert a[5][0] > 0 and not np.array_equal(a[4], s.box)  # moves were accepted


def test_partial_barostat_groups_take_the_reference_shaped_path(co, P):
    """group_idxs that leave atoms out (the reference accepts them: mol_utils.cpp checks range and uniqueness only): ungrouped atoms
    keep x while the box and their neighbours move, which the fast path's filter margin does not cover -- those barostats never take
    it, and their trajectories equal the fast-path-off run bit for bit."""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand()
    N = s.num_atoms
    groups = ts.molecule_groups(s)
    partial = groups[len(groups) // 2 :]

    def run(fast):
        with _Switches(co, merge=True, static_k=0, fast=fast):
            bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
            baro = MonteCarloBarostat(N, 50.0, 300.0, partial, 3, 19).impl(bps)
            ctxt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 2).impl(), bps, movers=[baro])
            xs, boxes = ctxt.multiple_steps(150, 3)
            return xs, boxes, baro.get_counters(), baro.get_attempt_paths()

    a, b = run(True), run(False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2] == b[2] and a[2][0] > 0
    assert a[3] == (50, 0) and b[3] == (50, 0)


def test_which_barostat_path_runs_in_which_composition(co, P):
    """`Context.get_barostat()` / the mover report the path: a single all-atom Nonbonded -> fast; the reference's composition -> fast
    from round 6 on (merged carrier), reference-shaped with merging off; an all-pairs potential over a SUBSET with no group -> never"""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s, n_lig = _system("config2")
    N = s.num_atoms
    groups = ts.molecule_groups(s)
    v0 = np.zeros_like(s.coords)

    def attempts_of(bps):
        baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 5, 3).impl(bps)
        ctxt = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, s.masses, 2).impl(), bps, movers=[baro])
        ctxt.multiple_steps(60, 0)
        assert ctxt.get_barostat() is not None
        return baro.get_attempt_paths()

    with _Switches(co, merge=True, static_k=0, fast=True):
        assert attempts_of([bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]) == (12, 12)
        assert attempts_of([bp.to_gpu(np.float32).bound_impl for bp in ts.rbfe_bound_potentials(s, n_lig)]) == (12, 12)
        host_only = ts.rbfe_bound_potentials(s, n_lig)[:-1]  # the group left out: the all-pairs potential covers the host only
        assert attempts_of([bp.to_gpu(np.float32).bound_impl for bp in host_only]) == (12, 0)
    with _Switches(co, merge=False, static_k=0, fast=True):
        assert attempts_of([bp.to_gpu(np.float32).bound_impl for bp in ts.rbfe_bound_potentials(s, n_lig)]) == (12, 0)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k", [("config2", 4608), ("config2", 0), ("config4", 0)])
def test_remembered_energies_are_the_recomputed_ones(co, P, which, static_k, precision):
    """Energy-only evaluations that gather for themselves are remembered on the device (csrc/engine.hpp: EnergyMemo): when every
    operand of the all-pairs items is unchanged the launch gets an empty item list and the sum is the remembered one.  A sequence of
    evaluations that exercises every way the operands can change (or not) -- the same frame under the same and other parameter sets
    (ligand only: the all-pairs part is skipped; host too: it is not), other frames, another box, a forces call in between (rewrites
    records without comparing: the memo must not be trusted), a Hilbert re-sort -- must give, call by call, the integers of a run
    with the memo switched off; for a single Nonbonded and for the reference's RBFE composition (merged carrier: the guest rows'
    items always run, in a launch of their own)."""
    from timemachine_amd import testsystems as ts

    s, n_lig = _system(which)
    N = s.num_atoms
    state = ts.rbfe_shaped_state(s, n_lig, env_charge_scale=0.9)
    rng = np.random.default_rng(8)
    frames = [s.coords + rng.normal(0.0, 0.002, s.coords.shape) for _ in range(3)]
    if precision == np.float32:
        frames = [f.astype(np.float32).astype(np.float64) for f in frames]
    box2 = s.box * 1.001
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
    sizes = [int(np.asarray(q).size) for _, q in state]
    off_group = sum(sizes[:-1])

    def window(lam, host_scale=1.0):
        p = flat.copy()
        g = p[off_group:].reshape(-1, 4)
        g[N - n_lig :, 0] *= 1.0 - 0.5 * lam
        g[N - n_lig :, 3] = lam * s.cutoff
        h = p[off_group - 4 * N : off_group].reshape(-1, 4)  # the host-host Nonbonded's block
        h[:, 0] *= host_scale
        return p

    nb_single = (P.Nonbonded(N, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff), None)
    summed = (P.SummedPotential([p for p, _ in state], [q for _, q in state]), None)

    def script(impl, params_of):
        out = []
        e = lambda x, prm, box=s.box: out.append(int(impl.execute_raw(x, prm, box, False, False, True)[2]))
        e(frames[0], params_of(0.0))
        e(frames[0], params_of(0.0))            # nothing changed
        e(frames[0], params_of(0.2))            # the ligand's parameters
        e(frames[0], params_of(0.2))
        e(frames[0], params_of(0.2, 0.97))      # the host's charges too
        e(frames[1], params_of(0.2, 0.97))      # another frame
        e(frames[1], params_of(0.4, 0.97))
        e(frames[1], params_of(0.4, 0.97), box2)  # another box, nothing else
        e(frames[1], params_of(0.4, 0.97), box2)
        impl.execute_raw(frames[2], params_of(0.0), s.box, True, False, False)  # a forces call: records rewritten, nothing compared
        e(frames[1], params_of(0.4, 0.97), box2)
        e(frames[2], params_of(0.0))
        for _ in range(101):                     # across the all-pairs potential's re-sort (every 100 calls)
            e(frames[2], params_of(0.1))
        e(frames[2], params_of(0.1))
        return out

    for pot, _ in (summed, nb_single):
        is_summed = pot is summed[0]

        def params_of(lam, host_scale=1.0):
            w = window(lam, host_scale)
            if is_summed:
                return w
            # one all-atom Nonbonded: host block with the ligand's rows taken from the group's block
            full = w[off_group - 4 * N : off_group].reshape(-1, 4).copy()
            full[N - n_lig :] = w[off_group:].reshape(-1, 4)[N - n_lig :]
            return full

        with _Switches(co, merge=True, static_k=static_k):
            before = co.debug_set_energy_memo(True)
            try:
                impl = pot.to_gpu(precision).unbound_impl
                with_memo = script(impl, params_of)
                evals, skipped = _all_pairs_of(impl).get_memo_stats()
                co.debug_set_energy_memo(False)
                impl2 = pot.to_gpu(precision).unbound_impl
                without = script(impl2, params_of)
                assert _all_pairs_of(impl2).get_memo_stats() == (0, 0)
            finally:
                co.debug_set_energy_memo(before)
        assert with_memo == without
        assert len(set(with_memo[:11])) >= (6 if is_summed else 5)  # the script does change the energy
        assert evals == len(with_memo)
        # skipped: call 2 (nothing changed), the repeats, and -- composition only -- the ligand-only changes
        expected_min = 100 if not is_summed else 103
        assert skipped >= expected_min, (evals, skipped)
        if not is_summed:
            assert skipped <= 106, (evals, skipped)


def test_hrex_energy_rows_with_the_memo_equal_rows_without(co, P):
    """compute_potential_matrix (fe/free_energy.py:1148-1200 through execute_batch_sparse) over three replicas' frames x the neighbouring
    windows of the RBFE composition: the same matrix with the memo on and off, and most evaluations skip the all-pairs launch"""
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts

    s, n_lig = _system("config2")
    N = s.num_atoms
    state = ts.rbfe_shaped_state(s, n_lig)
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
    off_group = flat.size - 4 * N
    n_states = 6
    params = np.stack([flat] * n_states)
    for k in range(n_states):
        g = params[k][off_group:].reshape(-1, 4)
        g[N - n_lig :, 3] = 0.1 * k * s.cutoff
        g[N - n_lig :, 0] *= 1.0 - 0.05 * k
    rng = np.random.default_rng(4)
    coords = np.stack([s.coords + rng.normal(0, 0.002, s.coords.shape) for _ in range(n_states)])
    boxes = np.stack([s.box] * n_states)
    rows = {}
    for memo in (True, False):
        before = co.debug_set_energy_memo(memo)
        try:
            impl = P.SummedPotential([p for p, _ in state], [q for _, q in state]).to_gpu(np.float32).unbound_impl
            rows[memo] = hrex.compute_potential_matrix(impl, coords, boxes, params, np.arange(n_states), max_delta_states=2)
            if memo:
                evals, skipped = _all_pairs_of(impl).get_memo_stats()
        finally:
            co.debug_set_energy_memo(before)
    np.testing.assert_array_equal(rows[True], rows[False])
    assert evals == np.isfinite(rows[True]).sum() and skipped == evals - n_states, (evals, skipped)  # one all-pairs launch per frame


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k", [("config2", 4608), ("config2", 0), ("config4", 0)])
def test_batches_over_parameter_sets_skip_the_list_kernels_and_keep_their_bits(co, P, which, static_k, precision):
    """execute_batch / execute_batch_sparse walk the parameter sets of one frame back to back (the reference orders its loops the same
    way, wrap_kernels.cpp:997-1001) and say so to their stateful children (csrc/engine.hpp: Potential::hint_same_frame): from a frame's
    second evaluation on no neighbor-list kernel is launched -- in every output form, whether or not the frame before needed a
    rebuild.  Frames far enough apart that every new frame rebuilds, parameter sets that differ in the ligand and in the host: the
    batch's integers equal those of the same batch with the hint ignored, entry by entry, and the skips are counted."""
    from timemachine_amd import testsystems as ts

    s, n_lig = _system(which)
    N = s.num_atoms
    state = ts.rbfe_shaped_state(s, n_lig, env_charge_scale=0.9)
    rng = np.random.default_rng(21)
    # (every frame a rigid 0.1 nm further along x than the one before: beyond padding / 2 for every atom, so every new frame rebuilds)
    frames = np.stack([s.coords + rng.normal(0.0, 0.002, s.coords.shape) + np.array([0.1 * k, 0.0, 0.0]) for k in range(4)])
    boxes = np.stack([s.box, s.box, s.box * 1.002, s.box])
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
    off_group = flat.size - 4 * N
    sets = np.stack([flat] * 4)
    for k in range(4):
        g = sets[k][off_group:].reshape(-1, 4)
        g[N - n_lig :, 0] *= 1.0 - 0.1 * k
        g[N - n_lig :, 3] = 0.1 * k * s.cutoff
    sets[3][off_group - 4 * N : off_group].reshape(-1, 4)[:, 0] *= 0.98  # the host's charges too
    ci = np.array([2, 0, 0, 1, 3, 3, 1, 0], dtype=np.uint32)
    pi = np.array([0, 1, 2, 3, 0, 1, 2, 3], dtype=np.uint32)
    summed = P.SummedPotential([p for p, _ in state], [q for _, q in state])
    forms = [(False, False, True), (True, False, False), (True, True, True)]

    def run(hint):
        before = co.debug_set_same_frame_hint(hint)
        try:
            impl = summed.to_gpu(precision).unbound_impl
            out = []
            for f in forms:
                out.append(impl.execute_batch(frames, sets, boxes, *f))
                out.append(impl.execute_batch_sparse(frames, sets, boxes, ci, pi, *f))
            return out, _all_pairs_of(impl).get_same_frame_skips()
        finally:
            co.debug_set_same_frame_hint(before)

    with _Switches(co, merge=True, static_k=static_k):
        with_hint, skips = run(True)
        without, skips_off = run(False)
    for a, b in zip(with_hint, without):
        for x, y in zip(a, b):
            if x is None:
                assert y is None
            else:
                np.testing.assert_array_equal(x, y)
    assert skips_off == 0
    if static_k == 0:  # (a static complete list launches no list kernel in the first place)
        # dense: 3 of 4 sets per frame x 4 frames; sparse: entries that share their frame with the entry before (sorted by frame): 4 of 8;
        # in the energy-only and the forces-only form the host-host potential's carrier is what is evaluated; the full form evaluates the two potentials
        assert skips >= 2 * (12 + 4), skips
    energies = with_hint[0][2]
    assert len(np.unique(energies)) == energies.size  # every (frame, set) pair is a different evaluation


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("barostat", [None, (4, 1.0, 3)])
def test_separate_producers_after_a_merged_stretch_do_not_find_their_old_pre_gather(co, P, barostat, precision):
    """A window whose interaction group loses and regains columns at run time (set_atom_idxs: what the reference's water-sampling and
    local moves do to their groups) steps  separate producers -> merged carrier -> separate producers.  While the carrier evaluates in
    their place the two potentials keep what an update kernel pre-gathered for them before the merged stretch -- behind the same
    coordinate pointers -- and must not use it when they become producers again (csrc/engine.hpp: carrier_took_over; found by
    tests/test_gpu_interleavings.py at config-5 size: attempts on the current list, unlike reference-shaped ones, do not happen to
    invalidate it).  Against the same calls with merging off; with and without a barostat."""
    s, n_lig = _system("config2")
    N = s.num_atoms
    host = np.arange(N - n_lig, dtype=np.int32)
    lig = np.arange(N - n_lig, N, dtype=np.int32)
    v0 = np.random.default_rng(4).normal(size=s.coords.shape) * 0.2

    def run(merge):
        with _Switches(co, merge=merge, static_k=0, fast=True):
            ctxt, bps, baro = _context(co, s, n_lig, precision, 0.1, v0, env_scale=0.9, barostat=barostat)
            group = bps[-1].get_potential()
            out = []
            for cols, n in ((host[:-9], 3), (host, 6), (host[:-12], 2), (host[:-9], 4), (host, 5)):
                group.set_atom_idxs(lig, cols)
                ctxt.multiple_steps(n, 0)
                out += [ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box()]
            return out, _host_all_pairs(bps).get_merged_stats()[0]

    merged, n_merged = run(True)
    plain, n_plain = run(False)
    assert n_plain == 0 and n_merged > 0
    for a, b in zip(merged, plain):
        np.testing.assert_array_equal(a, b)
    assert np.all(np.isfinite(merged[-3]))
