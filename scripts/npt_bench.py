"""ns/day of the DHFR-sized box with and without the Monte Carlo barostat (the reference benchmarks both:
tests/test_benchmark.py:517-518, barostat_interval in [0, 25]).  python scripts/npt_bench.py [f32|f64] [interval] [steps] [dhfr|config4|config2]"""
import sys
import time

import numpy as np

from timemachine_amd import potentials as P
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, custom_ops as co

prec = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
interval = int(sys.argv[2]) if len(sys.argv) > 2 else 25
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
co.set_device(0)
which = sys.argv[4] if len(sys.argv) > 4 else "dhfr"
s = {"dhfr": lambda: ts.dhfr_shaped_box(seed=2025, hmr=True, cutoff=1.2), "water": lambda: ts.dhfr_sized_water_box(seed=2025, hmr=True, cutoff=1.2), "config4": ts.config4_solvated_ligand, "config2": ts.small_solvated_ligand}[which]()
N = s.num_atoms
DT = 2.5e-3


def make_bps(p):
    bps = ts.bound_potentials(s, p, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]


eq = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 10.0, s.masses, 1).impl(), make_bps(np.float32))
eq.multiple_steps(3000, 0)
x, v = eq.get_x_t(), eq.get_v_t()
groups = ts.molecule_groups(s)
for label, iv in (("nvt", 0), ("npt", interval)):
    bps = make_bps(prec)
    movers = [MonteCarloBarostat(N, 1.0, 300.0, groups, iv, 7).impl(bps)] if iv > 0 else []
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, DT, 1.0, s.masses, 5).impl(), bps, movers=movers)
    ctxt.multiple_steps(1500, 0)
    ctxt.multiple_steps(steps, 0)
    ms = ctxt.last_multiple_steps_ms()
    t0 = time.perf_counter()
    ctxt.multiple_steps(steps, 0)
    host = time.perf_counter() - t0
    ms2 = ctxt.last_multiple_steps_ms()
    line = f"{label} {'f64' if prec == np.float64 else 'f32'} interval {iv}: {steps / (1e-3 * ms2) * 86400 * DT * 1e-3:8.1f} ns/day (device), {steps / host * 86400 * DT * 1e-3:8.1f} (host); us/step {1e3 * ms2 / steps:.2f}; box {np.diagonal(ctxt.get_box())[0]:.4f}"
    if movers:
        acc, att = movers[0].get_counters() if hasattr(movers[0], "get_counters") else (None, None)
        line += f"; accepted {acc}/{att}"
    print(line, flush=True)
