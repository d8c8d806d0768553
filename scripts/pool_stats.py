"""Per-workgroup (= per-CU pool) statistics of the tile kernel from a -DTM_TIMING build:
   TM_AMD_LIB=.../libtimemachine_amd_timing.so python scripts/pool_stats.py   (needs /tmp/ablate_frame.npy from ablate.py)"""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from timemachine_amd import potentials as P
from timemachine_amd import testsystems as ts
from timemachine_amd.lib import custom_ops as co

s = ts.dhfr_sized_water_box()
x = np.load("/tmp/ablate_frame.npy")
for prec, name, waves in ((np.float64, "f64", int(os.environ.get("WG_WAVES_F64", 12))), (np.float32, "f32", int(os.environ.get("WG_WAVES_F32", 10)))):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
    for _ in range(int(os.environ.get('WARM', 3))):
        nb.execute(x, s.nb_params, s.box, True, False, False)
    buf, cnt = nb.debug_timing(8192)
    t = buf.reshape(-1)[:cnt].reshape(-1, 8)
    items = (t[:, 4] & ((1 << 20) - 1)).astype(float)
    batches = (t[:, 5] & ((1 << 20) - 1)).astype(float)
    st = (t[:, 7] & 0xffffffff).astype(float)
    en = ((t[:, 7] >> 32) & 0xffffffff).astype(float)
    st0 = st.min()
    xcc = (t[:, 3] >> 56) & 0xf
    nwg = len(t) // waves
    W = lambda a: a[: nwg * waves].reshape(nwg, waves)
    wg_end = W(en).max(axis=1) - st0
    wg_items = W(items).sum(axis=1)
    wg_batches = W(batches).sum(axis=1)
    wg_xcc = W(xcc)[:, 0]
    cyc = W(t[:, 6].astype(float)).mean(axis=1)
    A = np.stack([wg_items, wg_batches, np.ones(nwg)], axis=1)
    coef, *_ = np.linalg.lstsq(A, wg_end, rcond=None)
    resid = wg_end - A @ coef
    print(f"{name}: {nwg} workgroups x {waves} waves | span {en.max() - st0:.0f} ticks | wg end pct " + " ".join(f"{np.percentile(wg_end, q):.0f}" for q in (0, 10, 50, 90, 100)))
    print(f"   items/wg pct " + " ".join(f"{np.percentile(wg_items, q):.0f}" for q in (0, 10, 50, 90, 100)) + " | batches/wg pct " + " ".join(f"{np.percentile(wg_batches, q):.0f}" for q in (0, 10, 50, 90, 100)))
    wave_end = W(en) - st0
    wave_start = W(st) - st0
    idle_in_wg = (wg_end[:, None] - wave_end).mean(axis=1)  # ticks a wave idles, on average, before its workgroup ends
    print(f"   inside a workgroup: waves start {wave_start.mean():.0f} +- {wave_start.std():.0f} ticks after the first; a wave is done {idle_in_wg.mean():.0f} ticks before its workgroup on average "
          f"(max over workgroups {idle_in_wg.max():.0f}); launch span {en.max() - st0:.0f}, mean workgroup end {wg_end.mean():.0f}, mean wave end {wave_end.mean():.0f}")
    xcc_wave = W(xcc)
    per_xcc_wave_end = [wave_end[xcc_wave == k].mean() for k in range(8)]
    print("   mean WAVE end per xcc: " + " ".join(f"{v:.0f}" for v in per_xcc_wave_end) + f" | largest {max(per_xcc_wave_end):.0f} vs launch span {en.max() - st0:.0f}"
          f" (what per-XCD dynamic pools could reach: every wave of an XCD ending together)")
    print(f"   fit end = {coef[0]:.1f}*items + {coef[1]:.2f}*batches + {coef[2]:.0f}; residual std {resid.std():.0f} ticks; corr(end,batches) {np.corrcoef(wg_end, wg_batches)[0,1]:.2f} corr(end,items) {np.corrcoef(wg_end, wg_items)[0,1]:.2f}")
    print("   mean end per xcc: " + " ".join(f"{wg_end[wg_xcc == k].mean():.0f}" for k in range(8)) + " | mean batches per xcc: " + " ".join(f"{wg_batches[wg_xcc == k].mean():.0f}" for k in range(8)))
    order = np.argsort(wg_end)
    print("   slowest wgs (idx, xcc, end, items, batches): " + " ".join(f"({i},{wg_xcc[i]},{wg_end[i]:.0f},{wg_items[i]:.0f},{wg_batches[i]:.0f})" for i in order[-6:]))
    print("   fastest wgs: " + " ".join(f"({i},{wg_xcc[i]},{wg_end[i]:.0f},{wg_items[i]:.0f},{wg_batches[i]:.0f})" for i in order[:6]))
