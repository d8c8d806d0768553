"""Hamiltonian replica exchange plumbing across GPUs.

What the reference does (single process; `timemachine/md/hrex.py`, `fe/free_energy.py:1148-1200,1537-1551`): after
every frame it evaluates the (replica, state) matrix of potential energies -- only states within `max_delta_states` of
each replica's current state, through `execute_batch_sparse` -- turns it into log-weights, and runs `n_states**3` seeded
neighbour-swap attempts on the state <-> replica permutation.

The MI355X form keeps every replica RESIDENT on its GPU (one process per GPU, replicas dealt round-robin): states -- i.e.
parameter vectors -- move, coordinates never do.  Per exchange step each rank evaluates the rows of the energy matrix
that belong to its replicas, one all_gather (RCCL over xGMI; a 24 x 24 f64 matrix is 4.6 KB, latency-bound) makes the
full matrix known to every rank, every rank runs the IDENTICAL seeded swap chain, and adopts the new state of its own
replicas with `BoundPotential.set_params`.

Random numbers: the reference draws pair indices and uniforms from `jax.random` (third party, absent here: parity
unpinned); this module uses numpy's PCG64.  The swap chain itself is deterministic given those draws and is restated
exactly (`run_neighbor_swaps` <-> `_run_neighbor_swaps`, md/hrex.py:51-130).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import parallel
from .constants import BOLTZ


def get_swap_attempts_per_iter_heuristic(n_states: int) -> int:
    """md/hrex.py:386-394"""
    return n_states**3


def neighbor_pairs(n_states: int) -> np.ndarray:
    """[(0,1), (1,2), ...] -- the allowed swaps of nearest-neighbour HREX (fe/free_energy.py run_sims_hrex)."""
    return np.stack([np.arange(n_states - 1), np.arange(1, n_states)], 1).astype(np.int64)


def run_neighbor_swaps(replica_idx_by_state, pairs, log_q_kl, pair_idxs, uniform_samples):
    """A batch of neighbour-swap Metropolis moves on the state -> replica permutation (md/hrex.py:51-130).

    log_q_kl[r, s] = log unnormalised probability of replica r in state s (-inf = not evaluated: never accepted).
    Returns (replica_idx_by_state, proposed[n_pairs], accepted[n_pairs]).  Runs in the native library
    (tm_hrex_run_neighbor_swaps); `run_neighbor_swaps_python` below is the same chain spelled out in numpy."""
    from .lib import custom_ops

    return custom_ops.hrex_run_neighbor_swaps(replica_idx_by_state, pairs, log_q_kl, pair_idxs, uniform_samples)


def run_neighbor_swaps_python(replica_idx_by_state, pairs, log_q_kl, pair_idxs, uniform_samples):
    """The chain of `run_neighbor_swaps`, statement by statement as md/hrex.py:91-121 has it (tests compare the two)."""
    perm = np.array(replica_idx_by_state, dtype=np.int64)
    pairs = np.asarray(pairs, dtype=np.int64)
    log_q_kl = np.asarray(log_q_kl, dtype=np.float64)
    proposed = np.zeros(len(pairs), dtype=np.uint32)
    accepted = np.zeros(len(pairs), dtype=np.uint32)
    with np.errstate(invalid="ignore", over="ignore"):
        for pair_idx, u in zip(np.asarray(pair_idxs, dtype=np.int64), np.asarray(uniform_samples, dtype=np.float64)):
            s_a, s_b = pairs[pair_idx]
            proposed[pair_idx] += 1
            r_a, r_b = perm[s_a], perm[s_b]
            log_q_before = log_q_kl[r_a, s_a] + log_q_kl[r_b, s_b]
            log_q_after = log_q_kl[r_a, s_b] + log_q_kl[r_b, s_a]
            log_q_diff = log_q_after - log_q_before
            acceptance_probability = np.exp(np.minimum(log_q_diff, 0.0))
            if u < acceptance_probability:  # False for NaN (inf - inf), as jnp comparisons are
                perm[s_a], perm[s_b] = r_b, r_a
                accepted[pair_idx] += 1
    return perm, proposed, accepted


def draw_swap_randomness(seed: int, n_pairs: int, n_swap_attempts: int) -> Tuple[np.ndarray, np.ndarray]:
    """(pair_idxs, uniforms) of one exchange step; identical on every rank for the same seed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, n_pairs, size=n_swap_attempts), rng.random(n_swap_attempts)


def sparse_batch_idxs(state_of_replica, n_states: int, max_delta_states: Optional[int], replicas: Optional[Sequence[int]] = None):
    """(coords_batch_idxs, params_batch_idxs) naming every (replica, state) with |state - current state| <= max_delta
    (fe/free_energy.py:1173-1179).  `replicas` restricts to a subset; the coords index is then the position in it."""
    state_of_replica = np.asarray(state_of_replica, dtype=np.int64)
    reps = np.arange(len(state_of_replica)) if replicas is None else np.asarray(list(replicas), dtype=np.int64)
    k = n_states if max_delta_states is None else int(max_delta_states)
    cand = state_of_replica[reps][:, None] + np.arange(-k, k + 1)[None, :]
    valid = np.nonzero((0 <= cand) & (cand < n_states))
    return valid[0].astype(np.uint32), cand[valid].astype(np.uint32)


def compute_potential_matrix(potential, coords, boxes, params_by_state, replica_idx_by_state, max_delta_states=None, replicas=None):
    """Rows of the (n_replicas, n_states) energy matrix for `replicas` (default: all), np.inf where not evaluated.

    potential: a `custom_ops.Potential` (unbound); coords [len(replicas), N, 3], boxes [len(replicas), 3, 3] in the order
    of `replicas`; params_by_state [n_states, ...].  fe/free_energy.py:1148-1200."""
    params_by_state = np.asarray(params_by_state, dtype=np.float64)
    n_states = params_by_state.shape[0]
    state_of_replica = np.argsort(np.asarray(replica_idx_by_state))
    reps = list(range(len(state_of_replica))) if replicas is None else list(replicas)
    ci, pi = sparse_batch_idxs(state_of_replica, n_states, max_delta_states, reps)
    rows = np.full((len(reps), n_states), np.inf)
    if len(ci):
        _, _, U = potential.execute_batch_sparse(np.asarray(coords), params_by_state, np.asarray(boxes), ci, pi, False, False, True)
        rows[ci, pi] = U
    return rows


def step_replicas(contexts, n_steps: int, group: int = 4):
    """n_steps of MD on every context of this rank.  Contexts that share a GPU are stepped `group` at a time through
    `custom_ops.multiple_steps_group` -- their steps interleaved on streams of their own, so that one replica's neighbor-list and
    integrator kernels (a quarter of a step, latency-bound) run underneath another's force kernel: two DHFR-sized replicas step at
    59.5 us each instead of 69.6, four at 55.0.  Trajectories are those of separate `multiple_steps` calls,
    bit for bit; a group whose contexts share a potential, integrator or mover object (refused by the grouped call) is stepped
    one context after the other instead.  The reference steps a device's windows one after the other
    (fe/free_energy.py:1537-1551)."""
    contexts = list(contexts)
    if not contexts:
        return  # a rank that owns no replica (more ranks than windows)
    grouped = getattr(contexts[0], "multiple_steps", None) is not None and group > 1 and len(contexts) > 1
    co = None
    if grouped:
        try:
            from .lib import custom_ops as co
        except ImportError:
            co = None
    if co is None or not all(isinstance(c, co.Context) for c in contexts):
        for c in contexts:  # stand-in contexts (bench.py --stub) or a single replica
            c.multiple_steps(n_steps, 0)
        return
    for k in range(0, len(contexts), group):
        chunk = contexts[k:k + group]
        try:
            co.multiple_steps_group(chunk, n_steps)
        except RuntimeError as e:
            # contexts that share device state (the reference's bind-one-potential-many-times pattern, a shared integrator):
            # the call refuses them BEFORE anything is enqueued -- step them one call after the other, which is always valid
            if "share a potential, integrator or mover" not in str(e):
                raise
            for c in chunk:
                c.multiple_steps(n_steps, 0)


def verify_and_sanitize_potential_matrix(U_kl, replica_idx_by_state, abs_energy_threshold: float = 1e9):
    """fe/free_energy.py:1203-1217: current-state energies must be finite and sane; NaN elsewhere becomes +inf."""
    U_kl = np.asarray(U_kl, dtype=np.float64)
    replica_energies = np.diagonal(U_kl[np.asarray(replica_idx_by_state)])
    assert np.all(np.isfinite(replica_energies)), "Replicas have non-finite energies"
    assert np.all(np.abs(replica_energies) < abs_energy_threshold), "Energies larger in magnitude than tolerated"
    return np.where(np.isnan(U_kl), np.inf, U_kl)


class DistributedHREX:
    """The state <-> replica bookkeeping of one HREX run, replicated on every rank.

    replica r lives on rank r % world for the whole run; `replica_idx_by_state[s]` says which replica currently samples
    state s.  `exchange()` is collective: every rank passes the energy rows of its own replicas."""

    def __init__(self, n_states: int, temperature: float, max_delta_states: Optional[int] = None, n_swap_attempts: Optional[int] = None,
                 world_size: int = 1, rank: int = 0):
        self.n_states = n_states
        self.kT = BOLTZ * temperature
        self.max_delta_states = max_delta_states
        self.n_swap_attempts = get_swap_attempts_per_iter_heuristic(n_states) if n_swap_attempts is None else n_swap_attempts
        self.world_size, self.rank = world_size, rank
        self.replica_idx_by_state = np.arange(n_states, dtype=np.int64)
        self.pairs = neighbor_pairs(n_states)
        self.replica_idx_by_state_by_iter: List[List[int]] = []
        self.fraction_accepted_by_pair_by_iter: List[List[Tuple[int, int]]] = []

    @property
    def local_replicas(self) -> List[int]:
        return parallel.windows_for_rank(self.n_states, self.world_size, self.rank)

    def state_of_replica(self) -> np.ndarray:
        return np.argsort(self.replica_idx_by_state)

    def exchange(self, local_rows: np.ndarray, seed: int):
        """local_rows [len(local_replicas), n_states]: energies (kJ/mol) of this rank's replicas, np.inf where not
        evaluated.  Returns the new state index of each local replica."""
        U_kl = parallel.gather_rows(self.local_replicas, local_rows, self.n_states, row_length=self.n_states)
        U_kl = verify_and_sanitize_potential_matrix(U_kl, self.replica_idx_by_state)
        log_q_kl = -U_kl / self.kT
        self.replica_idx_by_state_by_iter.append(self.replica_idx_by_state.tolist())
        pair_idxs, uniforms = draw_swap_randomness(seed, len(self.pairs), self.n_swap_attempts)
        self.replica_idx_by_state, proposed, accepted = run_neighbor_swaps(self.replica_idx_by_state, self.pairs, log_q_kl, pair_idxs, uniforms)
        self.fraction_accepted_by_pair_by_iter.append(list(zip(accepted.tolist(), proposed.tolist())))
        return self.state_of_replica()[self.local_replicas]
