"""Process CPU time (all threads, exact) over wall time for calls of several lengths: one context alone, four grouped.  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co
co.set_device(0)
s = ts.dhfr_sized_water_box()
def make(p):
    bps = ts.bound_potentials(s, p, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]
eq = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), make(np.float32))
eq.multiple_steps(1500, 0)
x, v = eq.get_x_t(), eq.get_v_t()
ctxts = [co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 5 + k).impl(), make(np.float64)) for k in range(4)]
out = []
for label, run, per in (("one", lambda n: ctxts[0].multiple_steps(n, 0), 1), ("four grouped", lambda n: co.multiple_steps_group(ctxts, n), 4)):
    run(200)
    for n in (1000, 2000, 4000, 8000, 2000):
        c0, w0 = time.process_time(), time.perf_counter()
        run(n)
        wall, cpu = time.perf_counter() - w0, time.process_time() - c0
        out.append(f"{label} {n}: {1e6 * wall / n / per:.1f} us/step, busy {cpu / wall:.2f}")
print(" | ".join(out), flush=True)
rows = []
for tid in os.listdir("/proc/self/task"):
    try:
        comm = open(f"/proc/self/task/{tid}/comm").read().strip()
        st = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
        rows.append((int(st[11]) + int(st[12]), comm, tid))
    except OSError:
        pass
print("threads by cpu ticks (10 ms):", sorted(rows, reverse=True)[:6], flush=True)
