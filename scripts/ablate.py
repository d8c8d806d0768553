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
        buf = np.zeros(8 * 8192, dtype=np.int64)
        cnt = ctypes.c_int(0)
        co._check(co._lib.tm_nonbonded_all_pairs_debug_timing(nb._h, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(buf.size), ctypes.byref(cnt)))
        t = buf[: cnt.value].reshape(-1, 8)
        if t[:, 6].sum() > 0:
            tot = t[:, 6].astype(float)
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


if __name__ == "__main__":
    worker() if len(sys.argv) > 1 else driver()
