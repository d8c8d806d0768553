"""Out-of-bounds write detector: run with TM_AMD_LIB pointing at a -DTM_GUARD build (build.build_variant("guard", ["TM_GUARD=1"])).
Every device buffer then sits between two guard zones; this drives MD in both precisions, energies / du_dp evaluations,
the barostat and an interaction-group state, and reports the zones that were written to."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, custom_ops as co


def check(tag):
    n = co.debug_check_guards()
    print(f"{tag}: guard violations = {n}", flush=True)
    return n


s = ts.dhfr_sized_water_box()
total = 0
for prec in (np.float32, np.float64):
    bps = [bp.to_gpu(prec).bound_impl for bp in ts.bound_potentials(s)]
    ctxt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), bps)
    ctxt.multiple_steps(400, 0)
    total += check(f"md {prec.__name__}")
    x = ctxt.get_x_t()
    for bp in bps:
        bp.execute(x, s.box, True, True)
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
    nb.execute(x, s.nb_params, s.box, True, True, True)
    nb.execute_batch(x[None], np.stack([s.nb_params, s.nb_params * 0.5]), s.box[None], True, True, True)
    total += check(f"evaluations {prec.__name__}")
small = ts.add_chain_ligand(ts.build_water_box(300, 3.0, seed=4), 16, lamb=0.3)
bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(small)]
ctxt = co.Context(small.coords, np.zeros_like(small.coords), small.box, LangevinIntegrator(300.0, 1.0e-3, 1.0, small.masses, 3).impl(), bps)
ctxt.multiple_steps(600, 0)
total += check("small system md")
# NPT (fused energy launches, pre-gathered before-energy), local MD (narrowed all-pairs + interaction group + restraints),
# and the SPLIT = 2 / 4 kernels of small systems in both precisions
mid = ts.config4_solvated_ligand()
for prec in (np.float32, np.float64):
    N = mid.num_atoms
    bps = [bp.to_gpu(prec).bound_impl for bp in ts.bound_potentials(mid)]
    groups = [list(range(3 * i, 3 * i + 3)) for i in range((N - 30) // 3)] + [list(range(N - 30, N))]
    baro = MonteCarloBarostat(N, 1.0, 300.0, groups, 5, 11).impl(bps)
    ctxt = co.Context(mid.coords, np.zeros_like(mid.coords), mid.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, mid.masses, 2).impl(), bps, movers=[baro])
    ctxt.multiple_steps(300, 0)
    total += check(f"npt {prec.__name__}")
    for fr in (True, False):
        ctxt = co.Context(mid.coords, np.zeros_like(mid.coords), mid.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, mid.masses, 2).impl(), bps)
        ctxt.setup_local_md(300.0, fr)
        ctxt.multiple_steps_local(100, np.arange(N - 30, N - 25, dtype=np.int32), radius=0.8, seed=3)
        ctxt.multiple_steps(50, 0)
        ref_atom, free = ctxt.local_md_last_selection()
        ctxt.multiple_steps_local_selection(50, ref_atom, free[free != ref_atom].astype(np.int32), radius=0.8)
    total += check(f"local md {prec.__name__}")
    for sysm in (ts.config1_water_cluster(3.0), ts.small_solvated_ligand()):
        b2 = [bp.to_gpu(prec).bound_impl for bp in ts.bound_potentials(sysm)]
        c2 = co.Context(sysm.coords, np.zeros_like(sysm.coords), sysm.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, sysm.masses, 2).impl(), b2)
        c2.multiple_steps(300, 0)
    total += check(f"split kernels {prec.__name__}")
# round 4: the row-block kernel on every forces-only launch (MD in both precisions, 4-D ligand state), and three contexts stepped
# together on the group streams
# (the row-block kernel lives in libraries built with -DTM_ROWBLOCK only: a guard build of that variant runs this part)
have_rb = co.debug_rowblock_available()
before = co.debug_set_rowblock_min_k(0) if have_rb else None
for prec in ((np.float32, np.float64) if have_rb else ()):
    for sysm in (s, small):
        bps = [bp.to_gpu(prec).bound_impl for bp in ts.bound_potentials(sysm)]
        ctxt = co.Context(sysm.coords, np.zeros_like(sysm.coords), sysm.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, sysm.masses, 1).impl(), bps)
        ctxt.multiple_steps(250, 0)
    total += check(f"row-block kernel {prec.__name__}")
if have_rb:
    co.debug_set_rowblock_min_k(before)
group = [co.Context(mid.coords, np.zeros_like(mid.coords), mid.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, mid.masses, 20 + k).impl(),
                    [bp.to_gpu(np.float64 if k else np.float32).bound_impl for bp in ts.bound_potentials(mid)]) for k in range(3)]
co.multiple_steps_group(group, 300)
total += check("group stepping")
# round 6: the reference's RBFE composition on the merged carrier (two record sets, guest rows' items, the holes of the merged order):
# MD and NPT in both precisions on the listed pipeline and on the static list, grouped; energy batches over frames x parameter sets
# (the energy memo, the same-frame hint, the f64 host entry points with device-side conversion)
for sysm, n_lig in ((ts.small_solvated_ligand(lamb=0.3), 20), (mid, 30)):
    Nm = sysm.num_atoms
    for prec in (np.float32, np.float64):
        for static_k in (0, 4608):
            k0 = co.debug_set_static_list_max_k(static_k)
            bps = [bp.to_gpu(prec).bound_impl for bp in ts.rbfe_bound_potentials(sysm, n_lig)]
            baro = MonteCarloBarostat(Nm, 1.0, 300.0, ts.molecule_groups(sysm), 5, 11).impl(bps)
            ctxt = co.Context(sysm.coords, np.zeros_like(sysm.coords), sysm.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, sysm.masses, 2).impl(), bps, movers=[baro])
            ctxt.multiple_steps(250, 0)
            co.debug_set_static_list_max_k(k0)
        state = ts.rbfe_shaped_state(sysm, n_lig)
        flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
        sets = np.stack([flat, flat, flat * 1.0])
        sets[1][flat.size - 4 * n_lig :].reshape(-1, 4)[:, 0] *= 0.9
        summed = P.SummedPotential([p for p, _ in state], [q for _, q in state]).to_gpu(prec).unbound_impl
        xs = np.stack([sysm.coords, ctxt.get_x_t()])
        bxs = np.stack([sysm.box, ctxt.get_box()])
        for flags in ((False, False, True), (True, False, False), (True, True, True)):
            summed.execute_batch(xs, sets, bxs, *flags)
            summed.execute_batch_sparse(xs, sets, bxs, np.array([1, 0, 1, 1], dtype=np.uint32), np.array([0, 1, 2, 1], dtype=np.uint32), *flags)
    total += check(f"rbfe composition, {Nm} atoms")
group = [co.Context(mid.coords, np.zeros_like(mid.coords), mid.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, mid.masses, 30 + k).impl(),
                    [bp.to_gpu(np.float32).bound_impl for bp in ts.rbfe_bound_potentials(mid, 30)]) for k in range(3)]
co.multiple_steps_group(group, 300)
total += check("group stepping, rbfe composition")
print("TOTAL", total)
sys.exit(1 if total else 0)
