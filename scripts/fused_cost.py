"""What do the piggy-backed bonded terms / exclusions cost the tile kernel?  MD on the DHFR-shaped box with subsets of its terms
(us per step and us per tile-kernel launch, f64 and f32).  GPU box only:  python scripts/fused_cost.py"""
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co  # noqa: E402

co.set_device(0)
full = ts.dhfr_shaped_box()
water = ts.dhfr_sized_water_box()
e0 = np.zeros((0, 2), np.int32)


def variants(s):
    full14 = np.all(s.scale_factors == 1.0, axis=1)
    yield "all terms", s
    yield "no torsions", dataclasses.replace(s, torsion_idxs=np.zeros((0, 4), np.int32), torsion_params=np.zeros((0, 3)))
    yield "no torsions, no angles", dataclasses.replace(s, torsion_idxs=np.zeros((0, 4), np.int32), torsion_params=np.zeros((0, 3)), angle_idxs=np.zeros((0, 3), np.int32), angle_params=np.zeros((0, 3)))
    yield "no 1-4 exclusions", dataclasses.replace(s, exclusion_idxs=s.exclusion_idxs[full14], scale_factors=s.scale_factors[full14])


def run(name, s, prec, x, v, steps=400):
    bps = ts.bound_potentials(s, prec, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    bp = summed.bind_params_list([b.params for b in bps]).to_gpu(prec).bound_impl
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, float(os.environ.get("FUSED_COST_DT", 2.5e-3)), 1.0, s.masses, 3).impl(), [bp])
    steps = int(os.environ.get("FUSED_COST_STEPS", steps))
    if steps > 50:
        ctxt.multiple_steps(300, 0)
    co.profile_reset()
    co.profile_set_enabled(True)
    ctxt.multiple_steps(steps, 0)
    ms, n = co.profile_read("nonbonded_tiles")
    co.profile_set_enabled(False)
    print(f"  {name:58s} {1e3 * ctxt.last_multiple_steps_ms() / steps:7.1f} us/step   tiles {1e3 * ms / max(n, 1):6.1f} us", flush=True)


for label, s in (("dhfr-shaped", full), ("water", water)):
    # equilibrate once (f32), reuse the frame for every variant
    frame = os.path.join(os.environ.get("FUSED_COST_FRAMES", "/tmp"), f"fused_cost_{label}.npz")
    if os.path.exists(frame):  # (an ablation build cannot equilibrate: reuse the product build's frame)
        x, v = np.load(frame)["x"], np.load(frame)["v"]
    else:
        x, v = s.coords.copy(), np.zeros_like(s.coords)
    for dt, fr, n in [] if os.path.exists(frame) else ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (2.5e-3, 1.0, 1000)):
        bps = [b.to_gpu(np.float32).bound_impl for b in ts.bound_potentials(s, np.float32, nblist_padding=0.18)]
        c = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, fr, s.masses, 5).impl(), bps)
        c.multiple_steps(n, 0)
        x, v = c.get_x_t(), c.get_v_t()
    np.savez(frame, x=x, v=v)
    for prec in (np.float64, np.float32):
        print(f"{label} {prec.__name__}")
        for name, sv in (variants(s) if label == "dhfr-shaped" else [("all terms", s)]):
            run(name, sv, prec, x, v)
