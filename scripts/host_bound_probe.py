"""Is a small composition's step host-bound?  Device time (HIP events), host wall time and process CPU time per step of one trajectory
of the RBFE composition (merged carrier) and of one all-atom Nonbonded at config-4 size.  usage: python scripts/host_bound_probe.py [f32|f64]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co
co.set_device(0)
prec = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
system, n_lig = ts.config4_solvated_ligand(0.3), 30
def make(single, packed=True):
    bound = ts.bound_potentials(system, prec, nblist_padding=0.18) if single else ts.rbfe_bound_potentials(system, n_lig, nblist_padding=0.18)
    if not packed:
        return [bp.to_gpu(prec).bound_impl for bp in bound]
    summed = P.SummedPotential([bp.potential for bp in bound], [bp.params for bp in bound])
    return [summed.bind_params_list([bp.params for bp in bound]).to_gpu(prec).bound_impl]
x, v = bench.equilibrate(co, LangevinIntegrator, system, lambda p: make(True), 7, 0.5, np.float32)
only = sys.argv[2] if len(sys.argv) > 2 else ""
for tag, single, packed in (("rbfe merged (one SummedPotential)", False, True), ("rbfe merged (eight bound potentials)", False, False), ("single Nonbonded (one SummedPotential)", True, True)):
    if only and only not in tag:
        continue
    ctxt = co.Context(x, v, system.box, LangevinIntegrator(bench.TEMPERATURE, bench.DT, bench.FRICTION, system.masses, 5).impl(), make(single, packed))
    ctxt.multiple_steps(500, 0)
    n = 3000
    c0, t0 = time.process_time(), time.perf_counter()
    ctxt.multiple_steps(n, 0)
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    print(f"{tag}: device {1e3 * ctxt.last_multiple_steps_ms() / n:.1f} us per step, host wall {1e6 * wall / n:.1f}, process CPU {1e6 * cpu / n:.1f}", flush=True)
