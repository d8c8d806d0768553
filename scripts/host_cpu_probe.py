"""Which host thread is busy while one Context steps: process CPU time and per-thread CPU times around multiple_steps.
python scripts/host_cpu_probe.py [steps=4000]"""
import os, sys, time
import numpy as np
import psutil

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
co.set_device(0)
s = {"dhfr": ts.dhfr_shaped_box, "water": ts.dhfr_sized_water_box, "config2": ts.small_solvated_ligand, "water900": lambda: ts.build_water_box(300, 3.0)}[os.environ.get("SYSTEM", "dhfr")]()
def make(p):
    bps = ts.bound_potentials(s, p, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]
x, v = s.coords.copy(), np.zeros_like(s.coords)
for dt, friction, n in ((0.1e-3, 100.0, 300), (0.5e-3, 50.0, 300), (1.0e-3, 10.0, 300), (2.5e-3, 1.0, 600)):
    c = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), make(np.float32))
    c.multiple_steps(n, 0)
    x, v = c.get_x_t(), c.get_v_t()
proc = psutil.Process()
def threads():
    return {t.id: t.user_time + t.system_time for t in proc.threads()}
for label, n_ctx in (("one context, multiple_steps", 1), ("one context, multiple_steps_group", -1), ("four contexts, multiple_steps_group", 4)):
    ctxts = [co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 100 + k).impl(), make(np.float64)) for k in range(abs(n_ctx))]
    run = (lambda n: ctxts[0].multiple_steps(n, 0)) if n_ctx == 1 else (lambda n: co.multiple_steps_group(ctxts, n))
    run(300)
    t_before, c0, w0 = threads(), time.process_time(), time.perf_counter()
    run(steps)
    wall, cpu = time.perf_counter() - w0, time.process_time() - c0
    t_after = threads()
    busy = sorted(((t_after[k] - t_before.get(k, 0.0)) / wall for k in t_after), reverse=True)[:4]
    print(f"{label:38s}: wall {1e6 * wall / steps / abs(n_ctx):6.1f} us per replica-step, CPUs busy {cpu / wall:.2f}; busiest threads {[round(b, 2) for b in busy]}", flush=True)
