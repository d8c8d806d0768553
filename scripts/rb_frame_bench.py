"""Tile-kernel time on a fixed equilibrated frame (forces-only launches through the host API, timed by the library's per-launch
profiler): usable with ablation builds whose forces are wrong.  First call (any library) writes /tmp/rb_frame.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

s = ts.dhfr_shaped_box()
if not os.path.exists("/tmp/rb_frame.npz"):
    x, v = s.coords.copy(), np.zeros_like(s.coords)
    for dt, friction, steps in ((0.1e-3, 100.0, 300), (0.5e-3, 50.0, 300), (1.0e-3, 10.0, 300), (2.5e-3, 1.0, 300)):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
        ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), bps)
        ctxt.multiple_steps(steps, 0)
        x, v = ctxt.get_x_t(), ctxt.get_v_t()
    np.savez("/tmp/rb_frame.npz", x=x)
x = np.load("/tmp/rb_frame.npz")["x"]
out = []
for min_k in (int(a) for a in os.environ.get("RB_MIN_KS", "0,2000000000").split(",")):
    co.debug_set_rowblock_min_k(min_k)
    for prec in (np.float64, np.float32):
        nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff, nblist_padding=0.18).to_gpu(prec).unbound_impl
        for _ in range(3):
            nb.execute(x, s.nb_params, s.box, True, False, False)
        co.profile_reset()
        co.profile_set_enabled(True)
        for _ in range(20):
            nb.execute(x, s.nb_params, s.box, True, False, False)
        ms, n = co.profile_read("nonbonded_tiles")
        co.profile_set_enabled(False)
        out.append(f"{'rowblock' if min_k == 0 else 'items'} {prec.__name__}: {1e3 * ms / max(n, 1):.1f} us per launch ({n} launches)")
print(os.path.basename(os.environ.get("TM_AMD_LIB", "default")), " | ".join(out))
