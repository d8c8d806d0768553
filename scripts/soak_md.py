import os, sys, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, custom_ops as co
co.set_device(0)
s = ts.dhfr_shaped_box(seed=2025, hmr=True)
def make_bps(p, padding=0.18):
    bps = ts.bound_potentials(s, p, nblist_padding=padding)
    summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
    return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]
x, v = bench.equilibrate(co, LangevinIntegrator, s, make_bps, 1234)
for prec in (np.float64, np.float32):
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 7).impl(), make_bps(prec))
    t0 = time.time()
    for chunk in range(4):
        ctxt.multiple_steps(50000, 0)
        xx, vv = ctxt.get_x_t(), ctxt.get_v_t()
        ke = 0.5 * (s.masses[:, None] * vv * vv).sum()
        T = 2 * ke / (3 * s.num_atoms * 0.0083144626)
        print(prec.__name__, "steps", (chunk + 1) * 50000, "T = %.1f K" % T, "finite", bool(np.all(np.isfinite(xx)) and np.all(np.isfinite(vv))), "max |v| %.2f" % np.abs(vv).max(), flush=True)
    print("  wall %.1f s for 200k steps" % (time.time() - t0))
# round 4: four replicas stepped together (custom_ops.multiple_steps_group), 100k steps each, both precisions; and the row-block
# kernel alone for 50k steps (f64)
for prec in (np.float64, np.float32):
    group = [co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 70 + k).impl(), make_bps(prec)) for k in range(4)]
    t0 = time.time()
    for chunk in range(2):
        co.multiple_steps_group(group, 50000)
        for k, c in enumerate(group):
            vv = c.get_v_t()
            T = (s.masses[:, None] * vv * vv).sum() / (3 * s.num_atoms * 0.0083144626)
            ok = bool(np.all(np.isfinite(c.get_x_t())) and np.all(np.isfinite(vv)))
            print(prec.__name__, "grouped replica", k, "steps", (chunk + 1) * 50000, "T = %.1f K" % T, "finite", ok, flush=True)
    print("  wall %.1f s for 4 x 100k steps (%.1f us per replica-step)" % (time.time() - t0, 1e6 * (time.time() - t0) / 4e5))
    del group
if co.debug_rowblock_available():  # (variant library libtimemachine_amd_rowblock.so only)
    co.debug_set_rowblock_min_k(0)
    ctxt = co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, 9).impl(), make_bps(np.float64))
    ctxt.multiple_steps(50000, 0)
    vv = ctxt.get_v_t()
    print("row-block kernel f64, 50k steps: T = %.1f K" % ((s.masses[:, None] * vv * vv).sum() / (3 * s.num_atoms * 0.0083144626)), "finite", bool(np.all(np.isfinite(ctxt.get_x_t()))), "; %.1f us per step" % (1e3 * ctxt.last_multiple_steps_ms() / 50000))
# round 5: the barostat's attempt on the current list over a long run -- NPT at 1 bar, an attempt every 25
# steps, 100k steps per precision alone, then the production shape (f32, four windows stepped together, a barostat in each)
from timemachine_amd.lib import MonteCarloBarostat
def npt_context(prec, seed):
    bps = make_bps(prec)
    baro = MonteCarloBarostat(s.num_atoms, 1.0, 300.0, ts.molecule_groups(s), 25, seed).impl(bps)
    return co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, seed).impl(), bps, movers=[baro]), baro
def report(tag, c, baro):
    vv, box = c.get_v_t(), c.get_box()
    T = (s.masses[:, None] * vv * vv).sum() / (3 * s.num_atoms * 0.0083144626)
    accepted, proposed = baro.get_counters()
    attempts, on_list = baro.get_attempt_paths()
    print(tag, "T = %.1f K" % T, "box %.4f nm" % box[0, 0], "finite", bool(np.all(np.isfinite(c.get_x_t())) and np.all(np.isfinite(vv))),
          "attempts %d (on the current list %d), accepted %.2f" % (attempts, on_list, accepted / max(proposed, 1)), flush=True)
for prec in (np.float64, np.float32):
    c, baro = npt_context(prec, 11)
    t0 = time.time()
    for chunk in range(4):
        c.multiple_steps(25000, 0)
        report("%s NPT steps %d" % (prec.__name__, (chunk + 1) * 25000), c, baro)
    print("  wall %.1f s for 100k steps (%.1f us per step)" % (time.time() - t0, 1e6 * (time.time() - t0) / 1e5))
group = [npt_context(np.float32, 20 + k) for k in range(4)]
t0 = time.time()
for chunk in range(2):
    co.multiple_steps_group([c for c, _ in group], 25000)
    for k, (c, baro) in enumerate(group):
        report("float32 NPT grouped replica %d steps %d" % (k, (chunk + 1) * 25000), c, baro)
print("  wall %.1f s for 4 x 50k steps (%.1f us per replica-step)" % (time.time() - t0, 1e6 * (time.time() - t0) / 2e5))
