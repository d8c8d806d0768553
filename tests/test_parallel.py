"""CPU tests of the replica-sharding path with a real 2-process gloo job (the GPU path is the same code over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_windows_for_rank_partitions_everything():
    from timemachine_amd.parallel import windows_for_rank

    for n, w in ((8, 8), (24, 8), (8, 2), (5, 3), (3, 8)):
        seen = sorted(k for r in range(w) for k in windows_for_rank(n, w, r))
        assert seen == list(range(n))
        sizes = [len(windows_for_rank(n, w, r)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    assert windows_for_rank(24, 8, 3) == [3, 11, 19]


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist

    from timemachine_amd import parallel

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    wins = parallel.windows_for_rank(5, world, rank)
    rows = np.array([[10.0 * k + l for l in range(5)] for k in wins])
    full = parallel.gather_rows(wins, rows, 5)
    tmax = parallel.max_over_ranks(1.0 + rank)
    parallel.barrier()
    q.put((rank, full, tmax))
    dist.destroy_process_group()


def test_gather_rows_world_size_2_gloo():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = np.array([[10.0 * k + l for l in range(5)] for k in range(5)])
    for rank, full, tmax in results:
        np.testing.assert_array_equal(full, expected)
        assert tmax == 2.0


def test_gather_rows_single_process():
    from timemachine_amd.parallel import gather_rows

    out = gather_rows([0, 1, 2], np.arange(6.0).reshape(3, 2), 3)
    np.testing.assert_array_equal(out, np.arange(6.0).reshape(3, 2))
