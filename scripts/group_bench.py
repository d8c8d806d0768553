"""Several replicas of the DHFR-shaped box on ONE GPU, stepped together (custom_ops.multiple_steps_group: the contexts' steps
interleaved on their own streams) against one replica alone.  python scripts/group_bench.py [f64|f32] [steps=2000]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

prec = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
which = sys.argv[3] if len(sys.argv) > 3 else "dhfr"
s = {"dhfr": ts.dhfr_shaped_box, "config2": ts.small_solvated_ligand, "config1": lambda: ts.config1_water_cluster(3.0), "config4": ts.config4_solvated_ligand}[which]()
counts = [int(a) for a in os.environ.get("GROUP_COUNTS", "1,2,3,4").split(",")]

def make(p):
    bps = ts.bound_potentials(s, p, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]

x, v = s.coords.copy(), np.zeros_like(s.coords)
for dt, friction, n in ((0.1e-3, 100.0, 300), (0.5e-3, 50.0, 300), (1.0e-3, 10.0, 300), (2.5e-3, 1.0, 600)):
    c = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), make(np.float32))
    c.multiple_steps(n, 0)
    x, v = c.get_x_t(), c.get_v_t()
import ctypes  # what the HIP runtime was told about hardware queues: exported by the native library itself when it is loaded (c_api.cpp)
_libc = ctypes.CDLL(None)
_libc.getenv.restype = ctypes.c_char_p
print("binding", getattr(co, "BINDING", "?"), "| GPU_MAX_HW_QUEUES as the runtime sees it:", _libc.getenv(b"GPU_MAX_HW_QUEUES"), "| in os.environ at start:", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for n_rep in counts:
    ctxts = [co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 100 + k).impl(), make(prec)) for k in range(n_rep)]
    co.multiple_steps_group(ctxts, 500)
    t0 = time.perf_counter()
    co.multiple_steps_group(ctxts, steps)
    wall = time.perf_counter() - t0
    ms = [c.last_multiple_steps_ms() for c in ctxts]
    agg = n_rep * steps / wall * 86400.0 * 2.5e-6
    print(f"{which} N={s.num_atoms}: {n_rep} replica(s) {prec.__name__}: wall {1e6 * wall / steps:7.2f} us per round of steps = {1e6 * wall / steps / n_rep:6.2f} us per replica-step; aggregate {agg:7.1f} ns/day; device ms per context {[round(m, 1) for m in ms]}", flush=True)
    assert all(np.all(np.isfinite(c.get_x_t())) for c in ctxts)
