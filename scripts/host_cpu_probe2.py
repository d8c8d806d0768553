"""Host CPU against the LENGTH of a multiple_steps call (DHFR-shaped box, f64): per call, the busiest threads' CPU time over the
call's wall time.  Without the run-ahead bound (TM_AMD_SPIN_WAIT=1) calls of 4 000 steps and every call after them keep the
enqueueing thread 0.9 busy inside the HIP runtime; with it every length stays at 0.3-0.5.  GPU box only."""
import os, sys, time
import numpy as np, psutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co
co.set_device(0)
s = ts.dhfr_sized_water_box()
def make(p):
    bps = ts.bound_potentials(s, p, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]
eq = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), make(np.float32))
eq.multiple_steps(2000, 0)
ctxt = co.Context(eq.get_x_t(), eq.get_v_t(), s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 5).impl(), make(np.float64))
proc = psutil.Process()
for n in (300, 1000, 2000, 4000, 8000, 2000):
    t0 = {t.id: t.user_time + t.system_time for t in proc.threads()}
    w0 = time.perf_counter()
    ctxt.multiple_steps(n, 0)
    wall = time.perf_counter() - w0
    t1 = {t.id: t.user_time + t.system_time for t in proc.threads()}
    busy = sorted(((t1[k] - t0.get(k, 0.0)) / wall for k in t1), reverse=True)[:3]
    print(n, "steps: wall us/step", round(1e6 * wall / n, 2), "busiest threads", [round(b, 2) for b in busy], flush=True)
