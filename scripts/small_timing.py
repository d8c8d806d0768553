"""Where a near-empty tile-kernel launch spends its time (a -DTM_TIMING build):
   TM_AMD_LIB=.../libtimemachine_amd_timing.so python scripts/small_timing.py [f32|f64] [config1|config4]"""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from timemachine_amd import potentials as P
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import custom_ops as co

prec = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
which = sys.argv[2] if len(sys.argv) > 2 else "config1"
s = ts.config1_water_cluster(3.0) if which == "config1" else ts.config4_solvated_ligand()
waves = 12 if prec == np.float64 else 10
nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
for _ in range(4):
    nb.execute(s.coords, s.nb_params, s.box, True, False, False)
buf, cnt = nb.debug_timing(8192)
t = buf.reshape(-1)[:cnt].reshape(-1, 8)
items = t[:, 4] & ((1 << 20) - 1)
st = (t[:, 7] & 0xFFFFFFFF).astype(float)
en = ((t[:, 7] >> 32) & 0xFFFFFFFF).astype(float)
st0 = st.min()
print(f"{which} {'f64' if prec == np.float64 else 'f32'}: {len(t)} waves, {int(items.sum())} items; kernel span (first wave start .. last wave end) {(en.max() - st0) * 0.01:.2f} us")
print(f"  wave start offsets us: pct " + " ".join(f"{np.percentile(st - st0, q) * 0.01:.2f}" for q in (0, 10, 50, 90, 100)))
busy = items > 0
print(f"  waves with items: {busy.sum()}; their life (start..end) us pct " + " ".join(f"{np.percentile((en - st)[busy], q) * 0.01:.2f}" for q in (0, 50, 100)))
print(f"  idle waves' life us pct " + " ".join(f"{np.percentile((en - st)[~busy], q) * 0.01:.2f}" for q in (0, 50, 100)))
cyc = t[:, 6].astype(float)
for name, col in (("setup (stage A: item -> LDS)", 0), ("phase 1", 1), ("phase 2", 2)):
    v = t[:, col][busy].astype(float) / np.maximum(items[busy], 1)
    print(f"  {name}: cycles per item pct " + " ".join(f"{np.percentile(v, q):.0f}" for q in (0, 50, 100)))
fl = (t[:, 3] & ((1 << 40) - 1))[busy].astype(float) / np.maximum(items[busy], 1)
print(f"  flush: cycles per item pct " + " ".join(f"{np.percentile(fl, q):.0f}" for q in (0, 50, 100)))
print(f"  total cycles of busy waves pct " + " ".join(f"{np.percentile(cyc[busy], q):.0f}" for q in (0, 50, 100)) + f" ; idle waves " + " ".join(f"{np.percentile(cyc[~busy], q):.0f}" for q in (0, 50, 100)))
acc = cyc[busy] - (t[:, 0] + t[:, 1] + t[:, 2])[busy] - (t[:, 3] & ((1 << 40) - 1))[busy]
print(f"  busy waves: cycles outside the item stages (prologue, first fetch, barrier, epilogue) pct " + " ".join(f"{np.percentile(acc, q):.0f}" for q in (0, 50, 100)))
