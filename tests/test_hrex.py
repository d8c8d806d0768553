"""CPU tests of the HREX plumbing (BASELINE config 5 shape: 24 windows cycling over 8 ranks):
the swap chain against the move-object formulation, the sparse (replica, state) index sets, and a real 2-process gloo
exchange in which both ranks must reach the same permutation."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swap_chain_matches_move_formulation():
    from oracle import hrex as ohrex
    from timemachine_amd import hrex

    rng = np.random.default_rng(3)
    for n_states in (2, 5, 24):
        pairs = hrex.neighbor_pairs(n_states)
        assert pairs.shape == (n_states - 1, 2) and hrex.get_swap_attempts_per_iter_heuristic(n_states) == n_states**3
        log_q = rng.normal(scale=2.0, size=(n_states, n_states))
        log_q[rng.uniform(size=log_q.shape) < 0.2] = -np.inf  # states outside max_delta_states
        np.fill_diagonal(log_q, rng.normal(size=n_states))
        perm0 = rng.permutation(n_states)
        n_attempts = 400
        pair_idxs = rng.integers(0, len(pairs), n_attempts)
        uniforms = rng.random(n_attempts)
        perm, proposed, accepted = hrex.run_neighbor_swaps(perm0, pairs, log_q, pair_idxs, uniforms)
        ref_perm, ref_prop, ref_acc = ohrex.run_moves(perm0.tolist(), pairs.tolist(), log_q.tolist(), pair_idxs.tolist(), uniforms.tolist())
        assert perm.tolist() == ref_perm and proposed.tolist() == ref_prop and accepted.tolist() == ref_acc
        assert sorted(perm.tolist()) == list(range(n_states)) and proposed.sum() == n_attempts
    # equal weights: every proposal accepted; forbidden (-inf) target states: never
    flat = np.zeros((4, 4))
    _, prop, acc = hrex.run_neighbor_swaps(np.arange(4), hrex.neighbor_pairs(4), flat, [0, 1, 2, 0], [0.999, 0.5, 0.0, 0.3])
    assert acc.tolist() == prop.tolist()
    blocked = np.full((4, 4), -np.inf)
    np.fill_diagonal(blocked, 0.0)
    perm, _, acc = hrex.run_neighbor_swaps(np.arange(4), hrex.neighbor_pairs(4), blocked, [0, 1, 2] * 5, np.zeros(15))
    assert acc.sum() == 0 and perm.tolist() == [0, 1, 2, 3]


def test_sparse_batch_idxs_follow_the_reference_formula():
    """fe/free_energy.py:1173-1179: neighbours within max_delta_states of each replica's CURRENT state."""
    from timemachine_amd import hrex

    replica_idx_by_state = np.array([2, 0, 3, 1, 4])  # state -> replica
    state_of_replica = np.argsort(replica_idx_by_state)
    ci, pi = hrex.sparse_batch_idxs(state_of_replica, 5, 1)
    expect = {(r, s) for r in range(5) for s in range(5) if abs(s - state_of_replica[r]) <= 1}
    assert set(zip(ci.tolist(), pi.tolist())) == expect and ci.dtype == np.uint32 and pi.dtype == np.uint32
    ci, pi = hrex.sparse_batch_idxs(state_of_replica, 5, None)
    assert len(ci) == 25
    ci, pi = hrex.sparse_batch_idxs(state_of_replica, 5, 1, replicas=[1, 4])  # a rank's own replicas: local coords index
    assert set(zip(ci.tolist(), pi.tolist())) == {(0, s) for s in range(5) if abs(s - state_of_replica[1]) <= 1} | {
        (1, s) for s in range(5) if abs(s - state_of_replica[4]) <= 1
    }


def _energy_rows(dh, x_by_replica, centres):
    """synthetic 'potential': U(replica r in state s) = 50 (x_r - c_s)^2, evaluated only within max_delta_states"""
    from timemachine_amd import hrex

    reps = dh.local_replicas
    ci, pi = hrex.sparse_batch_idxs(dh.state_of_replica(), dh.n_states, dh.max_delta_states, reps)
    rows = np.full((len(reps), dh.n_states), np.inf)
    rows[ci, pi] = 50.0 * (x_by_replica[np.asarray(reps)[ci]] - centres[pi]) ** 2
    return rows


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from timemachine_amd import hrex

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n_states = 6
    dh = hrex.DistributedHREX(n_states, 300.0, max_delta_states=2, world_size=world, rank=rank)
    centres = np.linspace(0.0, 1.0, n_states)
    x = np.random.default_rng(11).normal(loc=centres, scale=0.15)  # identical on every rank; each uses only its own
    history = []
    for it in range(5):
        new_states = dh.exchange(_energy_rows(dh, x, centres), seed=100 + it)
        history.append((dh.replica_idx_by_state.tolist(), new_states.tolist()))
    q.put((rank, dh.local_replicas, history, dh.fraction_accepted_by_pair_by_iter))
    dist.destroy_process_group()


def test_distributed_exchange_world_size_2_gloo():
    import torch.multiprocessing as mp

    from timemachine_amd import hrex

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, reps0, hist0, frac0), (_, reps1, hist1, frac1) = results
    assert reps0 == [0, 2, 4] and reps1 == [1, 3, 5]
    assert [h[0] for h in hist0] == [h[0] for h in hist1] and frac0 == frac1  # same permutation on both ranks
    # the single-process run of the same protocol gives the same chain
    n_states = 6
    dh = hrex.DistributedHREX(n_states, 300.0, max_delta_states=2)
    centres = np.linspace(0.0, 1.0, n_states)
    x = np.random.default_rng(11).normal(loc=centres, scale=0.15)
    for it in range(5):
        dh.exchange(_energy_rows(dh, x, centres), seed=100 + it)
        assert dh.replica_idx_by_state.tolist() == hist0[it][0]
        state_of = dh.state_of_replica()
        assert state_of[reps0].tolist() == hist0[it][1] and state_of[reps1].tolist() == hist1[it][1]
    assert any(h[0] != list(range(n_states)) for h in hist0), "no swap was ever accepted"
    assert sum(a for it in frac0 for a, _ in it) > 0


def test_sanitize_rejects_broken_replicas():
    from timemachine_amd import hrex

    U = np.array([[1.0, np.nan], [np.inf, 2.0]])
    out = hrex.verify_and_sanitize_potential_matrix(U, [0, 1])
    assert np.isinf(out[0, 1]) and out[0, 0] == 1.0
    with pytest.raises(AssertionError, match="non-finite"):
        hrex.verify_and_sanitize_potential_matrix(np.array([[np.inf, 0.0], [0.0, 1.0]]), [0, 1])


def test_step_replicas_falls_back_to_sequential_calls_for_stand_in_contexts():
    """hrex.step_replicas steps real Contexts `group` at a time through custom_ops.multiple_steps_group; anything else (bench.py's
    stand-in contexts, a single replica) is stepped one call after the other.  Either way every context takes n_steps."""
    from timemachine_amd import hrex

    class Stand:
        def __init__(self):
            self.calls = []

        def multiple_steps(self, n, interval=0):
            self.calls.append((n, interval))

    cs = [Stand() for _ in range(5)]
    hrex.step_replicas(cs, 40, group=4)
    assert all(c.calls == [(40, 0)] for c in cs)
    hrex.step_replicas(cs[:1], 7, group=4)
    assert cs[0].calls == [(40, 0), (7, 0)]


def test_both_bindings_offer_group_stepping(any_binding):
    assert callable(any_binding.multiple_steps_group) and callable(any_binding.debug_set_rowblock_min_k)
