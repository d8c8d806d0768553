"""Which part of packing a composition into ONE SummedPotential costs the tile launch 2 us at config-4 size?  Device us per step with
different subsets of the eight bound potentials packed.  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co
co.set_device(0)
prec = np.float32
system, n_lig = ts.config4_solvated_ligand(0.3), 30
bound = ts.rbfe_bound_potentials(system, n_lig, nblist_padding=0.18)
print([type(b.potential).__name__ for b in bound])
def pack(bs):
    s = P.SummedPotential([bp.potential for bp in bs], [bp.params for bp in bs])
    return s.bind_params_list([bp.params for bp in bs]).to_gpu(prec).bound_impl
def single(p):
    b = ts.bound_potentials(system, p, nblist_padding=0.18)
    return [pack(b)] if p is prec else [P.SummedPotential([bp.potential for bp in b], [bp.params for bp in b]).bind_params_list([bp.params for bp in b]).to_gpu(p).bound_impl]
x, v = bench.equilibrate(co, LangevinIntegrator, system, single, 7, 0.5, np.float32)
variants = {"none packed": lambda: [b.to_gpu(prec).bound_impl for b in bound],
            "all packed": lambda: [pack(bound)],
            "bonded six packed": lambda: [pack(bound[:6])] + [b.to_gpu(prec).bound_impl for b in bound[6:]],
            "nonbonded two packed": lambda: [b.to_gpu(prec).bound_impl for b in bound[:6]] + [pack(bound[6:])],
            "first three packed": lambda: [pack(bound[:3])] + [b.to_gpu(prec).bound_impl for b in bound[3:]],
            "4..6 packed": lambda: [b.to_gpu(prec).bound_impl for b in bound[:3]] + [pack(bound[3:6])] + [b.to_gpu(prec).bound_impl for b in bound[6:]]}
for tag, mk in variants.items():
    ctxt = co.Context(x, v, system.box, LangevinIntegrator(bench.TEMPERATURE, bench.DT, bench.FRICTION, system.masses, 5).impl(), mk())
    ctxt.multiple_steps(500, 0)
    ctxt.multiple_steps(3000, 0)
    print(f"{tag}: {1e3 * ctxt.last_multiple_steps_ms() / 3000:.2f} us per step", flush=True)
