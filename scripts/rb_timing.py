"""Where a workgroup of the row-block kernel spends its time (-DTM_TIMING build: TM_AMD_LIB=.../libtimemachine_amd_timing.so).
Wave 0's cycles up to each barrier of a unit, summed over the workgroup's units; the workgroups' end times (100 MHz clock)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

s = ts.dhfr_shaped_box()
x, v = s.coords.copy(), np.zeros_like(s.coords)
for dt, friction, steps in ((0.1e-3, 100.0, 300), (0.5e-3, 50.0, 300), (1.0e-3, 10.0, 300), (2.5e-3, 1.0, 300)):
    bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), bps)
    ctxt.multiple_steps(steps, 0)
    x, v = ctxt.get_x_t(), ctxt.get_v_t()
co.debug_set_rowblock_min_k(0)
WAVES = int(os.environ.get("RB_WAVES", "8"))
names = ["prologue", "decode+requests+rows (-> B0)", "filter (-> B1)", "sort (-> B2)", "pops (-> B3)", "flush (-> B4)"]
for prec in (np.float64, np.float32):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff, nblist_padding=0.18).to_gpu(prec).unbound_impl
    for _ in range(4):
        nb.execute(x, s.nb_params, s.box, True, False, False)
    buf, cnt = nb.debug_timing(8192)
    t = buf.reshape(-1)[:cnt].reshape(-1, 8)
    t = t[(t[:, 6] & ((1 << 48) - 1)) > 0]
    total = t[:, 6] & ((1 << 48) - 1)
    units = t[:, 6] >> 48
    print(f"{prec.__name__}: {len(t)} workgroups, units per workgroup min/mean/max {units.min()}/{units.mean():.2f}/{units.max()}, total units {units.sum()}")
    print(f"  kernel cycles per workgroup: mean {total.mean():.0f}, min {total.min()}, max {total.max()}")
    trips, pops = int((t[:, 0] & 0xffffffff).sum()), int((t[:, 0] >> 32).sum())
    print(f"  pop trips {trips} ({trips / len(t):.0f} per workgroup), lane-pops {pops}: lane occupancy of a trip {pops / max(trips, 1) / 64:.3f}; cycles of the pops phase per trip of a wave: {t[:, 4].sum() / max(trips, 1) * WAVES:.0f}")
    loop_cyc, npieces = int((t[:, 1] & ((1 << 40) - 1)).sum()), int((t[:, 1] >> 40).sum())
    piece_cyc, wait3 = int((t[:, 3] & 0xffffffff).sum()), int((t[:, 3] >> 32).sum())
    print(f"  pieces {npieces} ({npieces / units.sum():.1f} per unit), trips per piece {trips / max(npieces, 1):.1f}; cycles per trip inside the loop {loop_cyc / max(trips, 1):.0f}; "
          f"piece overhead (records, ticket, column store) {(piece_cyc - loop_cyc) / max(npieces, 1):.0f} cycles per piece; wave-cycles waiting at B3 per unit and wave {wait3 / units.sum() / WAVES:.0f}")
    for k, n in [(2, names[2]), (4, names[4]), (5, names[5])]:
        print(f"  {n:34s} mean {t[:, k].mean():9.0f} cycles = {t[:, k].mean() / total.mean():.3f} of the workgroup's life; per unit {t[:, k].sum() / max(units.sum(), 1):8.0f}")
    b, e = t[:, 7] & 0xffffffff, (t[:, 7] >> 32) & 0xffffffff
    t0 = b.min()
    print(f"  workgroup start (us after the first): mean {(b - t0).mean() / 100:.2f} max {(b - t0).max() / 100:.2f};  end: min {(e - t0).min() / 100:.2f} mean {(e - t0).mean() / 100:.2f} max {(e - t0).max() / 100:.2f}")
