"""Replica placement and the end-of-run energy gather: one process per GPU, independent lambda windows.

The reference has no communication backend at all: multi-GPU is a ProcessPoolExecutor that pins one task per device
with CUDA_VISIBLE_DEVICES and returns results by pickle (timemachine/parallel/client.py:188-218).  Here the windows are
sharded over the ranks of a torch.distributed job (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" on CPU) and
the only inter-GPU traffic is the final gather of reduced potentials u[k, l] (a few KB): coordinates never move.
"""
from typing import List, Optional, Sequence

import numpy as np


def windows_for_rank(n_windows: int, world_size: int, rank: int) -> List[int]:
    """window k -> rank k mod world_size (mirrors CUDAPoolClient's round-robin device choice, client.py:203-218)."""
    return [k for k in range(n_windows) if k % world_size == rank]


def _dist():
    """torch.distributed when a job of more than one rank is running (a one-rank group counts only if
    TM_AMD_FORCE_COLLECTIVES is set: the GPU test of the RCCL code path on a single-GPU box), else None."""
    import os

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist if (dist.get_world_size() > 1 or os.environ.get("TM_AMD_FORCE_COLLECTIVES")) else None


def _device(dist):
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_rows(local_windows: Sequence[int], local_rows: np.ndarray, n_windows: int, row_length: Optional[int] = None) -> np.ndarray:
    """All ranks contribute rows u[k, :] for their windows k; every rank returns the full [n_windows, L] matrix.

    ONE collective per call when the row length is known (``row_length``, or any rank-local row to read it from and the
    round-robin placement of ``windows_for_rank``): an all_gather of a [ceil(n_windows / world), 1 + L] f64 block per rank
    (RCCL over xGMI on GPUs; a 24 x 24 matrix is 4.8 KB -- latency-bound).  The block is filled with one host-side numpy
    assignment and one host-to-device copy.  Ranks that cannot know L (no rows, no ``row_length``) trigger a small
    extra all_gather of (L, count) first -- every rank must then take the same branch, so pass ``row_length`` whenever
    some rank may be empty."""
    import torch

    local_windows = [int(k) for k in local_windows]
    local_rows = np.asarray(local_rows, dtype=np.float64)
    if row_length is not None:
        L = int(row_length)
    else:
        L = int(local_rows.reshape(len(local_windows), -1).shape[1]) if len(local_windows) else -1
    dist = _dist()
    if dist is None:
        L = max(L, 0)
        out = np.full((n_windows, L), np.nan)
        out[local_windows] = local_rows.reshape(len(local_windows), L)
        return out
    world = dist.get_world_size()
    dev = _device(dist)
    cap = -(-n_windows // world)
    # row_length must be passed on EVERY rank or on none (the two branches issue different collectives: a mix hangs the job),
    # and with it the placement must fit the round-robin capacity
    if row_length is not None and len(local_windows) > cap:
        raise ValueError(
            f"gather_rows(row_length=...): this rank holds {len(local_windows)} windows but the single-collective path assumes at most "
            f"ceil({n_windows} / {world}) = {cap} per rank (windows_for_rank placement); omit row_length for other placements")
    if row_length is None:
        # agree on L and on the per-rank capacity (a rank without rows does not know L; placement may not be round-robin)
        meta = torch.tensor([L, len(local_windows)], dtype=torch.int64, device=dev)
        metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(metas, meta)
        metas = metas.cpu().numpy().reshape(world, 2)
        L = int(metas[:, 0].max())
        cap = max(cap, int(metas[:, 1].max()))
    block = np.full((cap, 1 + L), np.nan)
    block[: len(local_windows), 0] = local_windows
    block[: len(local_windows), 1:] = local_rows.reshape(len(local_windows), L)
    mine = torch.from_numpy(block).to(dev)
    everyone = torch.empty((world * cap, 1 + L), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(everyone, mine)
    everyone = everyone.cpu().numpy()
    filled = np.isfinite(everyone[:, 0])
    out = np.full((n_windows, L), np.nan)
    out[everyone[filled, 0].astype(np.int64)] = everyone[filled, 1:]
    return out


def gather_objects(obj) -> list:
    """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable records: who ran what); [obj] without a job"""
    dist = _dist()
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float) -> float:
    import torch

    dist = _dist()
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float) -> float:
    import torch

    dist = _dist()
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    dist = _dist()
    if dist is not None:
        dist.barrier()
