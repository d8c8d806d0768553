"""How many of an item's 32 rows lie within the cutoff of the item's 64-column chunk's bounding box?  (What a row-broadcast
filter phase with wave-uniform row skipping could save.)  Runs on the GPU box: equilibrated DHFR-sized frame, Hilbert order,
the library's own neighbor list at cutoff + padding, 64-column chunks as the tile kernel forms them."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

s = ts.dhfr_sized_water_box()
x, v = s.coords.copy(), np.zeros_like(s.coords)
for dt, friction, steps in ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (2.5e-3, 1.0, 1000)):
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), bps)
    ctxt.multiple_steps(steps, 0)
    x, v = ctxt.get_x_t(), ctxt.get_v_t()
box = s.box
L = np.diagonal(box)
perm = co.HilbertSort(s.num_atoms).sort(x, box)
xs = x[perm]
# a few steps later the atoms have moved (the list is built with padding): emulate by listing at cutoff + padding
pad = float(sys.argv[1]) if len(sys.argv) > 1 else 0.18
nbl = co.Neighborlist_f64(s.num_atoms)
lists = nbl.get_nblist(xs, box, s.cutoff + pad)
rc = s.cutoff
tot_rows = used_rows = items = 0
hits = 0
used_hist = np.zeros(33, dtype=np.int64)
for rb, cols in enumerate(lists):
    cols = np.asarray(cols, dtype=np.int64)
    cols = cols[cols < s.num_atoms]
    rows = xs[rb * 32:(rb + 1) * 32]
    o = rows[0]
    for c0 in range(0, len(cols), 64):
        cj = xs[cols[c0:c0 + 64]]
        d = cj - o
        d -= L * np.rint(d / L)
        lo, hi = d.min(0), d.max(0)
        r = rows - o
        r -= L * np.rint(r / L)
        gap = np.maximum(0.0, np.maximum(lo - r, r - hi))
        near = (gap ** 2).sum(1) < rc * rc
        dd = r[:, None, :] - d[None, :, :]
        hits += int(((dd ** 2).sum(-1) < rc * rc).sum())
        used_rows += int(near.sum()); tot_rows += len(rows); items += 1
        used_hist[int(near.sum())] += 1
print(f"padding {pad}: items {items}, rows within cutoff of the chunk box: {used_rows / tot_rows:.3f} of all (mean {used_rows / items:.1f} of 32), pair slots hit {hits / (items * 2048):.3f}")
print("histogram of useful rows per item (0..32):", used_hist.tolist())
