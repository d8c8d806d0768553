"""Kernel-time ablation of k_nonbonded_tiles on a fixed, equilibrated DHFR-sized frame.
usage: python scripts/ablate.py            (driver: equilibrates once with the product library, then runs every variant)
       python scripts/ablate.py worker     (internal)"""
import glob
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
FRAME = "/tmp/ablate_frame.npy"


def worker():
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import custom_ops as co

    s = ts.dhfr_sized_water_box()
    x = np.load(FRAME)
    out = []
    for prec, name in ((np.float64, "f64"), (np.float32, "f32")):
        nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
        nb.execute(x, s.nb_params, s.box, True, False, False)
        co.profile_reset()
        co.profile_set_enabled(True)
        for _ in range(20):
            nb.execute(x, s.nb_params, s.box, True, False, False)
        ms, n = co.profile_read("nonbonded_tiles")
        co.profile_set_enabled(False)
        out.append(f"{name} {1e3 * ms / n:7.1f} us")
        import ctypes
        buf, cnt = nb.debug_timing(8192)
        t = buf.reshape(-1)[:cnt].reshape(-1, 8)
        if t[:, 6].sum() > 0:
            tot = t[:, 6].astype(float)
            bc = (t[:, 4] >> 20).astype(float); sa = (t[:, 5] >> 20).astype(float)
            t[:, 4] &= (1 << 20) - 1; t[:, 5] &= (1 << 20) - 1
            start = t[:, 7].astype(float) - t[:, 7].min(); end = start + tot
            bt = t[:, 5].astype(float)
            pct = lambda a: " ".join(f"{np.percentile(a, q):.0f}" for q in (0, 10, 50, 90, 100))
            xcd = " ".join(f"{tot[k::8].mean():.0f}" for k in range(8))
            out.append(f"[total pct {pct(tot)} | batches pct {pct(bt)} | corr {np.corrcoef(tot, bt)[0,1]:.2f} | per blockIdx%8 total {xcd} | "
                       f"cycles/batch-equiv {np.polyfit(bt, tot, 1)}]")
            idle = []
            xcc = (t[:, 3] >> 56) & 0xf; hwid = (t[:, 3] >> 40) & 0xffff
            t[:, 3] &= (1 << 40) - 1
            out.append(f"[xcc of first 16 blocks {xcc[:16].tolist()} | waves per xcc {np.bincount(xcc, minlength=8).tolist()} | "
                       f"distinct (xcc,se,sh,cu,simd) {len(set(zip(xcc.tolist(), ((hwid>>4)&0xfff).tolist())))}]")
            st_all = (t[:, 7] & 0xffffffff).astype(float); en_all = ((t[:, 7] >> 32) & 0xffffffff).astype(float)
            st0 = st_all.min(); span = en_all.max() - st0
            simd_key = xcc * 65536 + ((hwid >> 4) & 0xfff)
            simd_end = {}
            for kk, e in zip(simd_key.tolist(), en_all.tolist()):
                simd_end[kk] = max(simd_end.get(kk, 0), e)
            se = np.array(list(simd_end.values())) - st0
            out.append(f"[realtime (10 ns ticks): span {span:.0f} | wave end mean {(en_all - st0).mean():.0f} | latest start {(st_all - st0).max():.0f} | "
                       f"SIMD last-wave end pct {pct(se)} mean {se.mean():.0f}]")
            for k in range(8):
                sel = xcc == k
                st = st_all[sel]; en = en_all[sel]
                span_k = en.max() - st0
                idle.append(f"{span_k:.0f}/{(en - st0).mean() / span_k:.2f}")
            out.append(f"[per xcc: end of last wave / mean wave end as a fraction of it: {' '.join(idle)}]")
            out.append(f"[stageA {sa.mean():.0f} stageBC {bc.mean():.0f} start mean {start.mean():.0f} max {start.max():.0f} end mean {end.mean():.0f} max {end.max():.0f}]")
            out.append(
                f"[waves {len(t)} items/wave {t[:,4].mean():.2f} batches/wave {t[:,5].mean():.1f} | mean cycles(100MHz ticks?) total {tot.mean():.0f} max {tot.max():.0f} "
                f"setup {t[:,0].mean():.0f} p1 {t[:,1].mean():.0f} p2 {t[:,2].mean():.0f} flush {t[:,3].mean():.0f}]")
    print(os.path.basename(os.environ.get("TM_AMD_LIB", "product")), " | ".join(out), flush=True)


def driver():
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    s = ts.dhfr_sized_water_box()
    x, v = s.coords.copy(), np.zeros_like(s.coords)
    for dt, friction, steps in ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (2.5e-3, 1.0, 1000)):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
        ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), bps)
        ctxt.multiple_steps(steps, 0)
        x, v = ctxt.get_x_t(), ctxt.get_v_t()
    np.save(FRAME, x)
    libs = [None] + sorted(glob.glob(os.path.join(REPO, "timemachine_amd", "csrc", "libtimemachine_amd_*.so")))
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["TM_AMD_LIB"] = lib
        subprocess.run([sys.executable, __file__, "worker"], env=env)


def make_frame():
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    s = ts.dhfr_sized_water_box()
    x, v = s.coords.copy(), np.zeros_like(s.coords)
    for dt, friction, steps in ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (2.5e-3, 1.0, 1000)):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in ts.bound_potentials(s)]
        ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, dt, friction, s.masses, 1).impl(), bps)
        ctxt.multiple_steps(steps, 0)
        x, v = ctxt.get_x_t(), ctxt.get_v_t()
    np.save(FRAME, x)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "frame":
        make_frame()
    elif len(sys.argv) > 1:
        worker()
    else:
        driver()
