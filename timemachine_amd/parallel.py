"""Replica placement and the end-of-run energy gather: one process per GPU, independent lambda windows.

The reference has no communication backend at all: multi-GPU is a ProcessPoolExecutor that pins one task per device
with CUDA_VISIBLE_DEVICES and returns results by pickle (timemachine/parallel/client.py:188-218).  Here the windows are
sharded over the ranks of a torch.distributed job (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" on CPU) and
the only inter-GPU traffic is the final gather of reduced potentials u[k, l] (a few KB): coordinates never move.
"""
from typing import List, Sequence

import numpy as np


def windows_for_rank(n_windows: int, world_size: int, rank: int) -> List[int]:
    """window k -> rank k mod world_size (mirrors CUDAPoolClient's round-robin device choice, client.py:203-218)."""
    return [k for k in range(n_windows) if k % world_size == rank]


def gather_rows(local_windows: Sequence[int], local_rows: np.ndarray, n_windows: int) -> np.ndarray:
    """All ranks contribute rows u[k, :] for their windows k; every rank returns the full [n_windows, L] matrix.
    Uses all_gather on a padded buffer (RCCL all_gather over xGMI on GPUs; latency-bound at these sizes)."""
    import torch
    import torch.distributed as dist

    local_rows = np.asarray(local_rows, dtype=np.float64).reshape(len(local_windows), -1)
    L = local_rows.shape[1] if len(local_windows) else 0
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = np.full((n_windows, L), np.nan)
        out[list(local_windows)] = local_rows
        return out
    world = dist.get_world_size()
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    # agree on L and on the per-rank capacity
    meta = torch.tensor([L, len(local_windows)], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    L = int(max(m[0].item() for m in metas))
    cap = int(max(m[1].item() for m in metas))
    buf = torch.full((cap, L + 1), float("nan"), dtype=torch.float64, device=dev)
    for r, (k, row) in enumerate(zip(local_windows, local_rows)):
        buf[r, 0] = float(k)
        buf[r, 1:] = torch.as_tensor(row, dtype=torch.float64, device=dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = np.full((n_windows, L), np.nan)
    for b in bufs:
        b = b.cpu().numpy()
        for row in b:
            if np.isfinite(row[0]):
                out[int(row[0])] = row[1:]
    return out


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
