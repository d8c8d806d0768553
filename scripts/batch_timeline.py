"""The per-batch timeline of k_nonbonded_tiles (f64, forces only) from a -DTM_TIMING_BATCH library (TM_AMD_LIB): mean wall cycles one
wave spends between the stamps inside a 64-pair batch, four waves per SIMD competing.  GPU box only; the frame comes from
scripts/tile_ablate.py (FRAME)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402
from timemachine_amd.lib import custom_ops as co  # noqa: E402

co.set_device(0)
s = ts.dhfr_shaped_box()
x = np.load(os.environ.get("FRAME", "/tmp/tile_ablate_frame.npz"))["x"]
names = ["queue entry read + decode", "operand fetch (12 LDS reads + wait)", "displacement, d^2", "pair function (f64: table index + fetch + polynomial + LJ; f32: analytic erfc / exp / switch + LJ)",
         "3 products + magic-add conversion", "6 LDS atomics (issued + acknowledged)"]
for prec in (np.float64, np.float32):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff, nblist_padding=0.18).to_gpu(prec).unbound_impl
    for _ in range(5):
        nb.execute(x, s.nb_params, s.box, True, False, False)
    buf, cnt = nb.debug_timing(8192)
    t = buf.reshape(-1)[:cnt].reshape(-1, 8).astype(float)
    t = t[t[:, 7] > 0]
    n = t[:, 7].sum()
    tot = 0.0
    print(f"{prec.__name__}: {len(t)} waves, {n / len(t):.1f} batches per wave")
    for k, name in enumerate(names):
        c = t[:, k].sum() / n
        tot += c
        print(f"  {name:100s} {c:7.0f} cycles")
    print(f"  {'sum':100s} {tot:7.0f} cycles per batch (stamps serialise the wave's own LDS traffic: an upper bound on the unstamped batch)")
