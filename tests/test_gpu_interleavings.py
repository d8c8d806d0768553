"""GPU test (``-m gpu``): random interleavings of everything a caller can do with a window -- stepping (with a Monte Carlo barostat
inside), energy / force / fixed-point evaluations of the context's own bound potentials at the current and at other coordinates,
``set_x_t`` / ``set_box`` / ``set_params``, batches over frames x parameter sets on a second copy of the state -- with every fast
path of the engine switched ON against the same sequence with every one of them switched OFF.

The fast paths are caches and hand-overs with invalidation rules (csrc/engine.hpp): the merged carrier of the reference's RBFE
composition, the sorted hand-over to the integrator, the barostat's attempts on the current list, the energy memo, the same-frame
hint of the batch entry points, the re-use of a list across box rescalings.  Each has tests of its own
(tests/test_gpu_rbfe_composition.py, tests/test_gpu_barostat_cases.py); what those cannot show is that no ORDER of calls leaves one
of them holding state it should have dropped.  Every quantity here is a function of (coordinates, parameters, box) and of counter-
keyed random streams only, and sums are integers, so the two runs must agree bit for bit whatever the order.
Reference semantics: a custom_ops object has no hidden state a caller could observe (tests/test_context.py, tests/nonbonded/
test_consistency.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


class _AllSwitches:
    """every process-wide A/B switch of the engine set for a block and restored after it"""

    def __init__(self, co, fast, static_k):
        self.co, self.fast, self.static_k = co, fast, static_k

    def __enter__(self):
        co = self.co
        self.prev = (
            co.debug_set_merge_producers(self.fast),
            co.debug_set_static_list_max_k(self.static_k),
            co.debug_set_barostat_fast_path(self.fast),
            co.debug_set_energy_memo(self.fast),
            co.debug_set_same_frame_hint(self.fast),
        )

    def __exit__(self, *exc):
        co = self.co
        co.debug_set_merge_producers(self.prev[0])
        co.debug_set_static_list_max_k(self.prev[1])
        co.debug_set_barostat_fast_path(self.prev[2])
        co.debug_set_energy_memo(self.prev[3])
        co.debug_set_same_frame_hint(self.prev[4])


def _make_ops(seed, n_ops):
    """the op list of a run: (kind, integer argument, seed of the op's own random numbers)"""
    rng = np.random.default_rng(seed)
    kinds = ["steps"] * 5 + ["energy", "energy", "forces", "fixed", "elsewhere", "set_x", "set_box", "set_params", "restore_params",
                              "batch", "batch_sparse", "unbound", "velocities", "local", "bound_batch", "set_idxs", "restore_idxs"]
    ops, restore_at = [], -1
    for i in range(n_ops):
        k = kinds[rng.integers(len(kinds))]
        if i == restore_at:  # a changed atom set is put back a few calls later: most of a run is on the merged carrier
            k = "restore_idxs"
        elif k == "set_idxs":
            restore_at = i + int(rng.integers(2, 6))
        ops.append((k, int(rng.choice([1, 2, 3, 4, 6, 11, 27])), int(rng.integers(1 << 30))))
    return ops


def _run(co, P, which, precision, static_k, fast, ops):
    from test_gpu_rbfe_composition import _all_pairs_of, _context, _host_all_pairs, _system
    from timemachine_amd import testsystems as ts

    s, n_lig = _system(which)
    N = s.num_atoms
    v0 = np.random.default_rng(3).normal(size=s.coords.shape) * 0.2
    out, labels = [], []
    with _AllSwitches(co, fast, static_k):
        ctxt, bps, baro = _context(co, s, n_lig, precision, 0.1, v0, env_scale=0.9, barostat=(5, 1.0, 9))
        state = ts.rbfe_shaped_state(s, n_lig, env_charge_scale=0.9)
        prm_group = np.asarray(state[7][1], dtype=np.float64)
        prm_host = np.asarray(state[6][1], dtype=np.float64)
        flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
        off_group = flat.size - 4 * N
        sets = np.stack([flat] * 3)
        for k in range(3):
            g = sets[k][off_group:].reshape(-1, 4)
            g[N - n_lig :, 0] *= 1.0 - 0.1 * k
            g[N - n_lig :, 3] = 0.1 * k * s.cutoff
        summed = P.SummedPotential([p for p, _ in state], [q for _, q in state]).to_gpu(precision).unbound_impl
        for op_no, (kind, n, op_seed) in enumerate(ops):
            labels += [(op_no, kind, n)] * (len(out) - len(labels))
            rng = np.random.default_rng(op_seed)
            x, box = ctxt.get_x_t(), ctxt.get_box()
            if kind == "steps":
                xs, boxes = ctxt.multiple_steps(n, n if n > 2 else 0)
                out += [np.asarray(xs), np.asarray(boxes), ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box()]
            elif kind == "energy":
                out += [np.float64(bp.execute(x, box, False, True)[1]) for bp in bps]
            elif kind == "forces":
                out += [bp.execute(x, box, True, False)[0] for bp in bps]
            elif kind == "fixed":
                out += [np.asarray(v) for bp in bps[-3:] for v in bp.execute_fixed(x, box)]
            elif kind == "elsewhere":  # the context's own potentials asked about coordinates that are not the context's
                y = x + rng.normal(0.0, 0.0002 * n, x.shape)
                for bp in bps[-2:]:
                    du, u = bp.execute(y, box * (1.0 + 0.001 * n), True, True)
                    out += [du, np.float64(u)]
            elif kind == "set_x":
                ctxt.set_x_t(x + rng.normal(0.0, 0.0002 * n, x.shape))
            elif kind == "set_box":
                ctxt.set_box(box * (1.0 + 0.0005 * (n - 3)))
            elif kind == "set_params":
                which_bp = bps[-1] if n % 2 else bps[-2]
                base = prm_group if n % 2 else prm_host
                which_bp.set_params((base * np.array([1.0 - 0.02 * n, 1.0, 1.0, 1.0])).reshape(-1))
            elif kind == "restore_params":
                bps[-1].set_params(prm_group.reshape(-1))
                bps[-2].set_params(prm_host.reshape(-1))
            elif kind in ("batch", "batch_sparse"):
                frames = np.stack([x, x + rng.normal(0.0, 0.003, x.shape), x + np.array([0.12, 0.0, 0.0])])
                boxes = np.stack([box, box, box * 1.002])
                form = [(False, False, True), (True, False, False), (True, True, True), (True, False, True)][op_seed % 4]
                if kind == "batch":
                    res = summed.execute_batch(frames, sets, boxes, *form)
                else:
                    ci = np.array([2, 0, 0, 1, 1, 0], dtype=np.uint32)
                    pi = np.array([0, 1, 2, 2, 0, 1], dtype=np.uint32)
                    res = summed.execute_batch_sparse(frames, sets, boxes, ci, pi, *form)
                out += [np.asarray(r) for r in res if r is not None]
            elif kind == "unbound":
                res = summed.execute(x, sets[n % 3], box, True, n % 2 == 0, True)
                out += [np.asarray(r) for r in res if r is not None]
            elif kind == "velocities":
                out += [ctxt.get_v_t()]
            elif kind == "local":  # local MD around the ligand (context.cu:90-213): its own restraint + selection, no barostat
                xs, boxes = ctxt.multiple_steps_local(2 * n, np.arange(N - n_lig, N - n_lig + 6, dtype=np.int32), 0, 1.2, 1000.0, op_seed % 1000)
                out += [np.asarray(xs), ctxt.get_x_t(), ctxt.get_v_t()]
            elif kind in ("set_idxs", "restore_idxs"):
                # the group's atom sets changed at run time (nonbonded_interaction_group.cu:63-127 set_atom_idxs): the merged carrier
                # exists only while the group's columns ARE the all-pairs potential's atom set
                host = np.arange(N - n_lig, dtype=np.int32)
                lig = np.arange(N - n_lig, N, dtype=np.int32)
                group_impl = bps[-1].get_potential()
                if kind == "restore_idxs":
                    group_impl.set_atom_idxs(lig, host)
                elif n % 2 == 0:  # fewer rows: still mergeable
                    group_impl.set_atom_idxs(lig[: n_lig - 1 - n % 5], host)
                else:  # fewer columns: no longer the all-pairs potential's atom set, evaluated separately until restored
                    group_impl.set_atom_idxs(lig, host[: -3 * (1 + n % 7)])
            elif kind == "bound_batch":
                frames = np.stack([x, x + rng.normal(0.0, 0.002, x.shape)])
                for bp in bps[-2:]:
                    res = bp.execute_batch(frames, np.stack([box, box * 1.001]), n % 2 == 0, True)
                    out += [np.asarray(r) for r in res if r is not None]
        paths = baro.get_attempt_paths() + (_host_all_pairs(bps).get_merged_stats()[0], _all_pairs_of(summed).get_memo_stats()[1], _all_pairs_of(summed).get_same_frame_skips())
        out += [ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box(), np.asarray(baro.get_counters()), np.float64(baro.get_volume_scale_factor())]
    labels += [(len(ops), "end", 0)] * (len(out) - len(labels))
    return out, paths, labels


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("which,static_k,seed,n_ops", [("config2", 0, 1, 70), ("config2", 4608, 2, 70), ("config4", 0, 4, 50)])
def test_random_interleavings_with_every_fast_path_equal_the_plain_paths(co, which, static_k, seed, n_ops, precision):
    from timemachine_amd import potentials as P

    ops = _make_ops(seed, n_ops)
    fast, paths_fast, labels = _run(co, P, which, precision, static_k, True, ops)
    plain, paths_plain, _ = _run(co, P, which, precision, static_k, False, ops)
    assert len(fast) == len(plain) and len(fast) > n_ops
    for k, (a, b) in enumerate(zip(fast, plain)):
        np.testing.assert_array_equal(a, b, err_msg=f"record {k} of {len(fast)} (ops: {[o[0] for o in ops]})")
    # (a trajectory that blew up would make the comparison above one of NaNs; single energies may be NaN by the overflow rule -- parameter
    # set 0 of the batches switches the ligand fully on where it overlaps the water)
    assert all(np.all(np.isfinite(a)) for a in fast[-5:-2]), "the trajectory did not stay finite"
    energies = [np.asarray(a) for a, l in zip(fast, labels) if np.asarray(a).ndim <= 2 and np.asarray(a).size <= 9]
    assert sum(int(np.isfinite(e).sum()) for e in energies) > 0.7 * sum(e.size for e in energies)
    # the comparison is between two different sets of code paths, not one path twice: (attempts, attempts on the current list, merged
    # evaluations of the context's carrier, remembered energy evaluations and list launches skipped on the same-frame hint in the batches)
    assert paths_plain[0] == paths_fast[0] > 0 and paths_plain[1:] == (0, 0, 0, 0), (paths_fast, paths_plain)
    assert paths_fast[1] > 0, paths_fast  # (attempts on the current list of this composition ARE the merged carrier's; its own count starts again with every new atom set)
    energy_only_batches = sum(1 for k, _, sd in ops if k in ("batch", "batch_sparse") and sd % 4 == 0)
    assert energy_only_batches > 0 and paths_fast[3] >= energy_only_batches, (paths_fast, energy_only_batches)
    assert static_k or paths_fast[4] > 0, paths_fast


def _run_windows(co, P, precision, static_k, fast, ops):
    """three windows of the RBFE composition at different lambda: stepped together and alone, their interaction-group parameters
    swapped between them (HREX: states move, coordinates stay), energy matrices over their frames in between"""
    from test_gpu_rbfe_composition import _context, _host_all_pairs
    from timemachine_amd import hrex
    from timemachine_amd import testsystems as ts

    lambdas = [0.27, 0.3, 0.33]  # (neighbouring states: a swap must not put a coupled ligand on top of the water)
    out = []
    with _AllSwitches(co, fast, static_k):
        windows, group_params, states = [], [], []
        for lamb in lambdas:
            s = ts.small_solvated_ligand(lamb=lamb)
            v0 = np.random.default_rng(11).normal(size=s.coords.shape) * 0.2
            windows.append(_context(co, s, 20, precision, 0.1, v0, barostat=(10, 1.0, 5)))
            state = ts.rbfe_shaped_state(s, 20)
            states.append(state)
            group_params.append(np.asarray(state[7][1], dtype=np.float64).reshape(-1))
        s0 = ts.small_solvated_ligand(lamb=0.0)
        summed = P.SummedPotential([p for p, _ in states[0]], [q for _, q in states[0]]).to_gpu(precision).unbound_impl
        flats = np.stack([np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in st]) for st in states])
        holder = list(range(3))  # which window holds which state's parameters
        for kind, n, op_seed in ops:
            if kind in ("steps", "local", "bound_batch"):  # together
                hrex.step_replicas([w[0] for w in windows], 2 * n, group=3 if n % 2 else 2)
            elif kind in ("energy", "forces", "fixed", "velocities"):  # one of them alone
                windows[n % 3][0].multiple_steps(n, 0)
            elif kind in ("set_params", "restore_params", "set_box"):  # a swap of neighbouring states
                a = n % 2
                holder[a], holder[a + 1] = holder[a + 1], holder[a]
                for k in (a, a + 1):
                    windows[k][1][-1].set_params(group_params[holder[k]])
            elif kind in ("batch", "batch_sparse", "unbound", "elsewhere"):
                coords = np.stack([w[0].get_x_t() for w in windows])
                boxes = np.stack([w[0].get_box() for w in windows])
                out += [hrex.compute_potential_matrix(summed, coords, boxes, flats, np.arange(3), max_delta_states=1 + n % 2)]
            elif kind == "set_x":
                w = windows[n % 3][0]
                w.set_x_t(w.get_x_t() + np.random.default_rng(op_seed).normal(0.0, 0.0002 * n, s0.coords.shape))
            for ctxt, _, _ in windows:
                out += [ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box()]
        stats = tuple(w[2].get_attempt_paths() for w in windows) + (sum(_host_all_pairs(w[1]).get_merged_stats()[0] for w in windows),)
    return out, stats


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("static_k,seed", [(0, 4), (4608, 7)])
def test_random_interleavings_of_windows_that_share_the_gpu(co, static_k, seed, precision):
    """the same for three windows: grouped stepping (hrex.step_replicas), stepping alone, parameter swaps between windows, energy
    matrices over the windows' frames (fe/free_energy.py:1383-1560's loop in random order)"""
    from timemachine_amd import potentials as P

    ops = _make_ops(seed, 40)
    fast, stats_fast = _run_windows(co, P, precision, static_k, True, ops)
    plain, stats_plain = _run_windows(co, P, precision, static_k, False, ops)
    assert len(fast) == len(plain)
    for k, (a, b) in enumerate(zip(fast, plain)):
        np.testing.assert_array_equal(a, b, err_msg=f"record {k} of {len(fast)}")
    assert all(np.all(np.isfinite(a)) for a in fast[-9:])
    assert stats_plain[3] == 0 and stats_fast[3] > 0 and all(f[0] == p[0] > 0 and p[1] == 0 and f[1] > 0 for f, p in zip(stats_fast[:3], stats_plain[:3])), (stats_fast, stats_plain)


def _run_single(co, precision, static_k, plain, ops, packed):
    """the benchmark's composition -- bonded terms + ONE all-atom Nonbonded (fe/free_energy.py:614-657), a barostat every 4 steps.
    ``plain``: every step is a call of its own behind set_x_t / set_v_t / set_box of the values just read (which drop every hand-over
    between the potential and the integrator: csrc/integrator.hip, Context::set_x_t), attempts reference-shaped; otherwise the calls
    as a user makes them.  ``packed``: the state as ONE SummedPotential (how the reference's fe layer binds it)."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    s = ts.small_solvated_ligand(lamb=0.3)
    N = s.num_atoms
    v0 = np.random.default_rng(17).normal(size=s.coords.shape) * 0.2
    out = []
    with _AllSwitches(co, not plain, static_k):
        parts = ts.bound_potentials(s, nblist_padding=0.1)
        flat = np.concatenate([np.asarray(bp.params, dtype=np.float64).reshape(-1) for bp in parts])
        if packed:
            bound = [P.SummedPotential([bp.potential for bp in parts], [bp.params for bp in parts]).bind(flat)]
        else:
            bound = parts
        bps = [bp.to_gpu(precision).bound_impl for bp in bound]
        baro = MonteCarloBarostat(N, 1.0, 300.0, ts.molecule_groups(s), 4, 3).impl(bps)
        ctxt = co.Context(s.coords, v0, s.box, LangevinIntegrator(300.0, 1.5e-3, 1.0, s.masses, 5).impl(), bps, movers=[baro])
        nb_params = np.asarray(parts[-1].params, dtype=np.float64)
        for kind, n, op_seed in ops:
            rng = np.random.default_rng(op_seed)
            x, box = ctxt.get_x_t(), ctxt.get_box()
            if kind in ("steps", "local", "bound_batch"):
                if plain:
                    for _ in range(n):
                        ctxt.set_x_t(ctxt.get_x_t())
                        ctxt.set_v_t(ctxt.get_v_t())
                        ctxt.set_box(ctxt.get_box())
                        ctxt.step()
                else:
                    ctxt.multiple_steps(n, 0)
            elif kind in ("energy", "fixed"):
                out += [np.float64(bp.execute(x, box, False, True)[1]) for bp in bps]
            elif kind == "forces":
                out += [bp.execute(x, box, True, False)[0] for bp in bps]
            elif kind in ("elsewhere", "unbound"):
                for bp in bps:
                    du, u = bp.execute(x + rng.normal(0.0, 0.0003 * n, x.shape), box * (1.0 + 0.001 * n), True, True)
                    out += [du, np.float64(u)]
            elif kind == "set_x":
                ctxt.set_x_t(x + rng.normal(0.0, 0.0002 * n, x.shape))
            elif kind == "set_box":
                ctxt.set_box(box * (1.0 + 0.0005 * (n - 3)))
            elif kind == "velocities":
                ctxt.set_v_t(ctxt.get_v_t() * (1.0 - 0.01 * n))
            elif kind in ("set_params", "restore_params"):
                scale = np.array([1.0 - 0.01 * n if kind == "set_params" else 1.0, 1.0, 1.0, 1.0])
                if packed:  # the whole state's parameters replaced behind the one bound potential (an HREX state move)
                    bps[0].set_params(np.concatenate([flat[: flat.size - nb_params.size], (nb_params * scale).reshape(-1)]))
                else:
                    bps[-1].set_params((nb_params * scale).reshape(-1))
            elif kind in ("batch", "batch_sparse"):
                frames = np.stack([x, x + rng.normal(0.0, 0.002, x.shape)])
                for bp in bps[-1:]:
                    res = bp.execute_batch(frames, np.stack([box, box * 1.001]), n % 2 == 0, True)
                    out += [np.asarray(r) for r in res if r is not None]
            out += [ctxt.get_x_t(), ctxt.get_v_t(), ctxt.get_box()]
        out += [np.asarray(baro.get_counters()), np.float64(baro.get_volume_scale_factor())]
        paths = baro.get_attempt_paths()
    return out, paths


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("static_k,seed,packed", [(0, 7, False), (4608, 8, False), (0, 10, True)])
def test_random_interleavings_on_the_benchmark_composition(co, static_k, seed, packed, precision):
    """one all-atom Nonbonded: the sorted hand-over to the integrator, the slot-ordered update, bonded terms riding on the tile launch,
    barostat attempts on the current list -- against a run in which every step is a call of its own behind setters that drop all of it"""
    ops = _make_ops(seed, 60)
    fast, paths_fast = _run_single(co, precision, static_k, False, ops, packed)
    plain, paths_plain = _run_single(co, precision, static_k, True, ops, packed)
    assert len(fast) == len(plain)
    for k, (a, b) in enumerate(zip(fast, plain)):
        np.testing.assert_array_equal(a, b, err_msg=f"record {k} of {len(fast)}")
    assert all(np.all(np.isfinite(a)) for a in fast[-5:-2])
    assert paths_fast[0] == paths_plain[0] > 0 and paths_plain[1] == 0 and paths_fast[1] > 0, (paths_fast, paths_plain)
