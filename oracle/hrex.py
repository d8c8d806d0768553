"""TEST INFRASTRUCTURE ONLY.  The slow, move-object formulation of neighbour swaps (timemachine/md/hrex.py:25-48,
`NeighborSwapMove` under `MixtureOfMoves`) restated on plain lists, to check the array implementation against:
propose a swap of the replicas at states (s_a, s_b), accept with probability min(1, q(r_a, s_b) q(r_b, s_a) / (q(r_a, s_a) q(r_b, s_b)))."""
import numpy as np


def neighbor_swap_move(state, log_q, s_a, s_b, uniform):
    """One NeighborSwapMove: `state` is the list replica_idx_by_state; returns (new_state, accepted)."""
    proposed = list(state)
    proposed[s_a], proposed[s_b] = state[s_b], state[s_a]
    r_a, r_b = state[s_a], state[s_b]
    with np.errstate(invalid="ignore"):
        log_q_diff = log_q(r_a, s_b) + log_q(r_b, s_a) - log_q(r_a, s_a) - log_q(r_b, s_b)
        log_acceptance_probability = np.minimum(log_q_diff, 0.0)
    if uniform < np.exp(log_acceptance_probability):
        return proposed, True
    return list(state), False


def run_moves(state, pairs, log_q_kl, pair_idxs, uniforms):
    state = list(state)
    proposed = [0] * len(pairs)
    accepted = [0] * len(pairs)
    for k, u in zip(pair_idxs, uniforms):
        s_a, s_b = pairs[k]
        state, ok = neighbor_swap_move(state, lambda r, s: log_q_kl[r][s], int(s_a), int(s_b), u)
        proposed[k] += 1
        accepted[k] += int(ok)
    return state, proposed, accepted
