"""Where an HREX frame's energy-matrix time goes (bench.py --mode hrex: 24 windows x 31k atoms, max_delta_states 4): the
execute_batch_sparse call with all (replica, state) entries, with one entry per replica, and with one entry in total.  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import hrex, potentials as P, testsystems as ts
from timemachine_amd.lib import custom_ops as co
co.set_device(0)
system = ts.config5_complex_sized(0.0)
N, n_states = system.num_atoms, 24
lig = np.arange(system.num_water_atoms, N)
params_by_state = np.stack([system.nb_params] * n_states)
for k, lam in enumerate(np.linspace(0.0, 0.5, n_states)):
    params_by_state[k][lig, 3] = lam * system.cutoff
    params_by_state[k][lig, 0] *= 1.0 - 0.5 * lam
rng = np.random.default_rng(3)
coords = np.stack([system.coords + rng.normal(0, 0.002, system.coords.shape) for _ in range(n_states)])
boxes = np.stack([system.box] * n_states)
for prec in (np.float64, np.float32):
    unbound = P.Nonbonded(N, system.exclusion_idxs, system.scale_factors, system.beta, system.cutoff).to_gpu(prec).unbound_impl
    state_of_replica = np.arange(n_states)
    ci, pi = hrex.sparse_batch_idxs(state_of_replica, n_states, 4, list(range(n_states)))
    def call(ci, pi, reps=5):
        unbound.execute_batch_sparse(coords, params_by_state, boxes, ci, pi, False, False, True)
        t0 = time.perf_counter()
        for _ in range(reps):
            unbound.execute_batch_sparse(coords, params_by_state, boxes, ci, pi, False, False, True)
        return 1e3 * (time.perf_counter() - t0) / reps
    full = call(ci, pi)
    diag = call(np.arange(n_states, dtype=np.uint32), np.arange(n_states, dtype=np.uint32))
    one = call(np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint32))
    print(f"{prec.__name__}: {len(ci)} entries {full:.2f} ms | {n_states} entries (one per replica) {diag:.2f} ms | 1 entry {one:.2f} ms"
          f" -> staging ~{one:.2f} ms, first evaluation of a frame ~{1e3 * (diag - one) / (n_states - 1):.0f} us, further parameter sets ~{1e3 * (full - diag) / (len(ci) - n_states):.0f} us each", flush=True)
