"""us per tile-kernel launch on a fixed equilibrated frame of the DHFR-shaped box, forces only, for whatever library TM_AMD_LIB
names (ablation builds cannot integrate: the frame comes from the product build, FRAME=path.npz).  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co  # noqa: E402



def timing_line(nb, name):
    """a -DTM_TIMING library: mean per-wave cycle counters of the last launch (setup, phase 1, phase 2, flush, prefetch)"""
    buf, cnt = nb.debug_timing(8192)
    t = buf.reshape(-1)[:cnt].reshape(-1, 8)
    if not cnt or t[:, 6].sum() == 0:
        return None
    bc = (t[:, 4] >> 20).astype(float)
    sa = (t[:, 5] >> 20).astype(float)
    items = (t[:, 4] & ((1 << 20) - 1)).astype(float)
    batches = (t[:, 5] & ((1 << 20) - 1)).astype(float)
    flush = (t[:, 3] & ((1 << 40) - 1)).astype(float)
    tot = t[:, 6].astype(float)
    return (f"   timing {name}: waves {len(t)} items/wave {items.mean():.2f} batches/wave {batches.mean():.1f} | cycles: total {tot.mean():.0f} (max {tot.max():.0f}) "
            f"setup {t[:, 0].mean():.0f} p1 {t[:, 1].mean():.0f} p2 {t[:, 2].mean():.0f} flush {flush.mean():.0f} prefetch {bc.mean():.0f} stageA {sa.mean():.0f}"
            f" | per item: setup {t[:, 0].sum() / items.sum():.0f} p1 {t[:, 1].sum() / items.sum():.0f} flush {flush.sum() / items.sum():.0f} prefetch {bc.sum() / items.sum():.0f}; per batch p2 {t[:, 2].sum() / batches.sum():.0f}")

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
out, lines = [], []
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
    tl = timing_line(nb, prec.__name__)
    if tl:
        lines.append(tl)
print(os.path.basename(os.environ.get("TM_AMD_LIB", "product")), " | ".join(out), flush=True)
for tl in lines:
    print(tl, flush=True)
