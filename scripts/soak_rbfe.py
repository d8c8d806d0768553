"""Round 6 soak: the reference's RBFE state composition (HostGuestSystem: testsystems.rbfe_bound_potentials) on the merged carrier at
config-5 size (31 378 atoms) -- NPT at 1 bar with an attempt every 25 steps, 200k steps per precision alone, then four f32 windows
stepped together (100k steps each, a barostat in each), with an energy-matrix style evaluation (execute_batch_sparse over the
windows' frames x neighbouring parameter sets: the energy memo and the same-frame hint) between the chunks.  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from timemachine_amd import hrex, potentials as P, testsystems as ts
from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, custom_ops as co

co.set_device(0)
n_lig = 40
s = ts.config5_complex_sized(0.3)
N = s.num_atoms
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000

def packed(prec, lamb_scale=1.0):
    bound = ts.rbfe_bound_potentials(s, n_lig, nblist_padding=0.18)
    summed = P.SummedPotential([bp.potential for bp in bound], [bp.params for bp in bound])
    return [summed.bind_params_list([bp.params for bp in bound]).to_gpu(prec).bound_impl]

x, v = bench.equilibrate(co, LangevinIntegrator, s, lambda p: packed(p), 1234, 0.5, np.float32)

def all_pairs_of(impl):
    if type(impl).__name__.startswith("NonbondedAllPairs"):
        return impl
    for c in impl.get_potentials() if hasattr(impl, "get_potentials") else []:
        r = all_pairs_of(c)
        if r is not None:
            return r
    return None

def context(prec, seed):
    bps = packed(prec)
    baro = MonteCarloBarostat(N, 1.0, 300.0, ts.molecule_groups(s), 25, seed).impl(bps)
    return co.Context(x, v, s.box, LangevinIntegrator(300.0, 2.5e-3, 1.0, s.masses, seed).impl(), bps, movers=[baro]), baro, bps

def report(tag, c, baro, bps):
    vv, box = c.get_v_t(), c.get_box()
    T = (s.masses[:, None] * vv * vv).sum() / (3 * N * 0.0083144626)
    accepted, proposed = baro.get_counters()
    attempts, on_list = baro.get_attempt_paths()
    merged = all_pairs_of(bps[0].get_potential()).get_merged_stats()
    print(tag, "T = %.1f K" % T, "box %.4f nm" % box[0, 0], "finite", bool(np.all(np.isfinite(c.get_x_t())) and np.all(np.isfinite(vv))),
          "attempts %d (on the current list %d), accepted %.2f; merged evaluations %d, list builds %d" % (attempts, on_list, accepted / max(proposed, 1), merged[0], merged[2]), flush=True)

for prec in (np.float64, np.float32):
    c, baro, bps = context(prec, 11)
    t0 = time.time()
    for chunk in range(4):
        c.multiple_steps(steps // 4, 0)
        report(f"{prec.__name__} NPT/25 steps {(chunk + 1) * (steps // 4)}", c, baro, bps)
    print("  wall %.1f s for %d steps (%.1f us per step)" % (time.time() - t0, steps, 1e6 * (time.time() - t0) / steps), flush=True)

# four f32 windows stepped together, an energy matrix between the chunks (the production shape of the reference's HREX loop)
state = ts.rbfe_shaped_state(s, n_lig, nblist_padding=0.18)
flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
off_group = flat.size - 4 * N
n_w = 4
params = np.stack([flat] * n_w)
for k in range(n_w):
    g = params[k][off_group:].reshape(-1, 4)
    g[N - n_lig:, 3] = 0.1 * k * s.cutoff
    g[N - n_lig:, 0] *= 1.0 - 0.05 * k
matrix_impl = P.SummedPotential([p for p, _ in state], [q for _, q in state]).to_gpu(np.float32).unbound_impl
group = [context(np.float32, 70 + k) for k in range(n_w)]
t0 = time.time()
for chunk in range(4):
    co.multiple_steps_group([g[0] for g in group], steps // 8)
    coords = np.stack([g[0].get_x_t() for g in group])
    boxes = np.stack([g[0].get_box() for g in group])
    u = hrex.compute_potential_matrix(matrix_impl, coords, boxes, params, np.arange(n_w), max_delta_states=2)
    before = co.debug_set_energy_memo(False)
    hint = co.debug_set_same_frame_hint(False)
    u_plain = hrex.compute_potential_matrix(matrix_impl, coords, boxes, params, np.arange(n_w), max_delta_states=2)
    co.debug_set_energy_memo(before)
    co.debug_set_same_frame_hint(hint)
    print("energy matrix after chunk", chunk + 1, "equal with the memo / hint off:", bool(np.array_equal(u, u_plain, equal_nan=True)), "finite entries", int(np.isfinite(u).sum()), flush=True)
    for k, (c, baro, bps) in enumerate(group):
        report(f"f32 grouped window {k} steps {(chunk + 1) * (steps // 8)}", c, baro, bps)
print("  wall %.1f s for 4 x %d steps" % (time.time() - t0, steps // 2))
ev, sk = all_pairs_of(matrix_impl).get_memo_stats()
print("matrix potential: memo evaluations %d, all-pairs launch skipped in %d; list kernels skipped on the same-frame hint: %d" % (ev, sk, all_pairs_of(matrix_impl).get_same_frame_skips()))
