"""Neighbor-list geometry model.  TEST INFRASTRUCTURE ONLY.

* ``block_bounds``: cpp/src/kernels/k_neighborlist.cuh:11-116 -- per 32-atom block, a running
  min/max that re-images each atom around the current centre (round-half-even, atoms visited in the
  lane order 1,2,...,31,0).  The reference's own numpy model is tests/test_nblist.py:28-56.
* ``brute_force_ixn_list``: tests/test_nblist.py:117-140 -- per row block, the set of column atoms
  (index >= block start) within ``cutoff`` of at least one row atom.  This is what ``get_nblist`` must
  return as a set per block (tests/test_nblist.py:180-186).
"""
import numpy as np

TILE = 32


def block_bounds(coords, box, block_size=TILE, real=np.float64):
    coords = np.asarray(coords, dtype=np.float64)
    b = np.diagonal(np.asarray(box, dtype=np.float64)).astype(real)
    inv_b = (real(1) / b).astype(real)
    N = coords.shape[0]
    nb = (N + block_size - 1) // block_size
    ctr = np.zeros((nb, 3), dtype=real)
    ext = np.zeros((nb, 3), dtype=real)
    half = real(0.5)
    for t in range(nb):
        blk = coords[t * block_size : min((t + 1) * block_size, N)].astype(real)
        lo = blk[0].copy()
        hi = blk[0].copy()
        order = list(range(1, len(blk))) + [0]
        for k in order:
            p = blk[k]
            c = half * (hi + lo)
            img = (p - b * np.rint((p - c) * inv_b)).astype(real)
            lo = np.minimum(lo, img)
            hi = np.maximum(hi, img)
        ctr[t] = half * (hi + lo)
        ext[t] = half * (hi - lo)
    return ctr.astype(np.float64), ext.astype(np.float64)


def _min_image(d, b):
    return d - b * np.floor(d / b + 0.5)  # timemachine/potentials/jax_utils.py:37-44


def brute_force_ixn_list(coords, box, cutoff, block_size=TILE):
    coords = np.asarray(coords, dtype=np.float64)
    b = np.diagonal(np.asarray(box, dtype=np.float64))
    N = coords.shape[0]
    nb = (N + block_size - 1) // block_size
    out = []
    for r in range(nb):
        r0, r1 = r * block_size, min((r + 1) * block_size, N)
        d = _min_image(coords[r0:r1, None, :] - coords[None, :, :], b)
        dij = np.linalg.norm(d, axis=-1)
        dij[:, :r0] = cutoff
        out.append(np.nonzero(np.any(dij < cutoff, axis=0))[0].tolist())
    return out


def brute_force_ixn_list_rows(coords, box, cutoff, row_idxs, block_size=TILE):
    """Row subset vs complement columns, tests/test_nblist.py:143-177."""
    coords = np.asarray(coords, dtype=np.float64)
    b = np.diagonal(np.asarray(box, dtype=np.float64))
    N = coords.shape[0]
    row_idxs = np.asarray(row_idxs)
    col_idxs = np.delete(np.arange(N), row_idxs)
    rows = coords[row_idxs]
    nb = (len(rows) + block_size - 1) // block_size
    out = []
    for r in range(nb):
        blk = rows[r * block_size : (r + 1) * block_size]
        d = _min_image(blk[:, None, :] - coords[col_idxs][None, :, :], b)
        dij = np.linalg.norm(d, axis=-1)
        out.append(col_idxs[np.nonzero(np.any(dij < cutoff, axis=0))[0]].tolist())
    return out


def tile_count(ixn_list, block_size=TILE):
    return sum((len(l) + block_size - 1) // block_size for l in ixn_list)
