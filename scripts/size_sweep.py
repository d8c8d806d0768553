"""us per MD step against system size (one Context, forces-only MD, padding 0.18): where the launch-latency floor is.
python scripts/size_sweep.py [f32|f64]"""
import sys

import numpy as np

from timemachine_amd import potentials as P
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

prec = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
co.set_device(0)
systems = [
    ("config1 256", ts.config1_water_cluster(3.0)),
    ("water 900", ts.build_water_box(300, 3.0)),
    ("config2 2.3k", ts.small_solvated_ligand()),
    ("config4 6.4k", ts.config4_solvated_ligand()),
    ("water 12k", ts.build_water_box(4000, 4.93)),
    ("dhfr 23.6k", ts.dhfr_sized_water_box()),
]
import os
if os.environ.get("SWEEP_SMALL"):
    systems = systems[:4] + [("water 3.6k", ts.build_water_box(1200, 3.3)), ("water 9k", ts.build_water_box(3000, 4.48))]
for name, s in systems:
    def make(p):
        bps = ts.bound_potentials(s, p, nblist_padding=0.18)
        summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
        return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]
    eq = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), make(np.float32))
    eq.multiple_steps(2000, 0)
    ctxt = co.Context(eq.get_x_t(), eq.get_v_t(), s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 5).impl(), make(prec))
    ctxt.multiple_steps(1000, 0)
    import time
    import psutil
    _proc = psutil.Process()
    th0 = {t.id: t.user_time + t.system_time for t in _proc.threads()}
    c0, w0 = time.process_time(), time.perf_counter()
    ctxt.multiple_steps(4000, 0)
    th1 = {t.id: t.user_time + t.system_time for t in _proc.threads()}
    busiest = sorted(((th1[k] - th0.get(k, 0.0)) / (time.perf_counter() - w0) for k in th1), reverse=True)[:3]
    ms = ctxt.last_multiple_steps_ms()
    ctxt.get_x_t()  # (waits for the device)
    cpu_us, wall_us = 1e6 * (time.process_time() - c0) / 4000, 1e6 * (time.perf_counter() - w0) / 4000
    # host columns: wall clock of the call per step (== the device's when the host keeps up) and process CPU time per step (launch
    # thread + HIP runtime helpers): a step whose host cost exceeds its device time is HOST-bound
    print(f"{name:14s} N={s.num_atoms:6d}  {1e3 * ms / 4000:7.2f} us/step  {4000 / (1e-3 * ms) * 86400 * 2.5e-6:9.1f} ns/day   host: wall {wall_us:6.2f} us/step, cpu {cpu_us:6.2f} us/step, busiest threads {[round(b, 2) for b in busiest]}", flush=True)
