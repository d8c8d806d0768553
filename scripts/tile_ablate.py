"""us per tile-kernel launch on a fixed equilibrated frame of the DHFR-shaped box, forces only, for whatever library TM_AMD_LIB
names (ablation builds cannot integrate: the frame comes from the product build, FRAME=path.npz).  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co  # noqa: E402

co.set_device(0)
s = ts.dhfr_shaped_box() if os.environ.get("WORKLOAD", "dhfr") == "dhfr" else ts.dhfr_sized_water_box()
frame = os.environ.get("FRAME", "/tmp/tile_ablate_frame.npz")
if os.path.exists(frame):
    x = np.load(frame)["x"]
else:
    x, v = s.coords.copy(), np.zeros_like(s.coords)
    for dt, fr, n in ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (2.5e-3, 1.0, 1000)):
        bps = [b.to_gpu(np.float32).bound_impl for b in ts.bound_potentials(s, np.float32, nblist_padding=0.18)]
        c = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, fr, s.masses, 5).impl(), bps)
        c.multiple_steps(n, 0)
        x, v = c.get_x_t(), c.get_v_t()
    np.savez(frame, x=x)
out = []
for prec in (np.float64, np.float32):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff, nblist_padding=0.18).to_gpu(prec).unbound_impl
    for _ in range(5):
        nb.execute(x, s.nb_params, s.box, True, False, False)
    co.profile_reset()
    co.profile_set_enabled(True)
    for _ in range(int(os.environ.get("REPS", 40))):
        nb.execute(x, s.nb_params, s.box, True, False, False)
    ms, n = co.profile_read("nonbonded_tiles")
    co.profile_set_enabled(False)
    out.append(f"{prec.__name__} {1e3 * ms / n:6.1f} us")
print(os.path.basename(os.environ.get("TM_AMD_LIB", "product")), " | ".join(out), flush=True)
