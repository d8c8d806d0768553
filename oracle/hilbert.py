"""Hilbert-curve ordering model.  TEST INFRASTRUCTURE ONLY.  Bit-exact (integer/index work).

Follows cpp/src/hilbert_sort.cu:13-47 (LUT: bin (i,j,k) -> hilbert_c2i(3, 8, {i,j,k})),
cpp/src/kernels/k_hilbert.cu:9-54 (key from coordinates) and hilbert_sort.cu:69-80 (stable radix
sort by key -> permutation).  The curve itself is the Butz algorithm as implemented by the vendored
third-party code cpp/src/vendored/hilbert.cpp:196-237 (Doug Moore, Rice University, 1998-2000);
``c2i_3d`` below is an independent restatement for nDims=3, pinned bit-for-bit against that C file
compiled into oracle/_ref/ (oracle/Makefile, tests/test_oracle_hilbert.py).
"""
import numpy as np

HILBERT_GRID_DIM = 128  # cpp/src/kernels/k_hilbert.cuh:6
HILBERT_N_BITS = 8  # k_hilbert.cuh:9


def c2i_3d(c0, c1, c2, nbits=HILBERT_N_BITS):
    """Vectorised Hilbert index of integer coordinates (arrays), nDims = 3.

    Per bit-plane b (MSB first) form the 3-bit group g_b = (c2_b c1_b c0_b); xor with the group of
    the plane above; undo the running (flip, rotation) state; append; update the state from the
    emitted digit: rotation += 1 + ffs(digit) (mod 3), flip = 1 << old rotation.  Finally xor with
    the constant 0b100100...100 pattern and Gray-decode (prefix xor from the top)."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    shape = c0.shape
    index = np.zeros(shape, dtype=np.uint64)
    rot = np.zeros(shape, dtype=np.uint64)
    flip = np.zeros(shape, dtype=np.uint64)
    prev = np.zeros(shape, dtype=np.uint64)
    one, three, seven = np.uint64(1), np.uint64(3), np.uint64(7)
    for b in range(nbits - 1, -1, -1):
        sb = np.uint64(b)
        g = ((c0 >> sb) & one) | (((c1 >> sb) & one) << one) | (((c2 >> sb) & one) << np.uint64(2))
        t = g ^ prev ^ flip
        prev = g
        digit = ((t >> rot) | (t << (three - rot))) & seven  # rotate right by rot within 3 bits
        index = (index << three) | digit
        flip = one << rot
        low = digit & (~digit + one) & three  # lowest set bit, restricted to the low nDims-1 bits
        inc = np.where(low == 0, 0, np.where(low == 1, 1, 2)).astype(np.uint64)
        rot = (rot + one + inc) % three
    nd = 3 * nbits
    pattern = 0
    for k in range(nbits):
        pattern |= 1 << (3 * k)
    index ^= np.uint64(pattern >> 1)
    d = 1
    while d < nd:
        index ^= index >> np.uint64(d)
        d *= 2
    return index


def build_lut():
    """bin_to_idx[i*128*128 + j*128 + k] (uint32), hilbert_sort.cu:18-31."""
    g = np.arange(HILBERT_GRID_DIM, dtype=np.uint64)
    i, j, k = np.meshgrid(g, g, g, indexing="ij")
    return c2i_3d(i.ravel(), j.ravel(), k.ravel()).astype(np.uint32)


_LUT = None


def lut():
    global _LUT
    if _LUT is None:
        _LUT = build_lut()
    return _LUT


def keys(coords, box, atom_idxs=None):
    """k_hilbert.cu:9-54: image into the home box with f64 floor; bin width = max box edge / 127."""
    coords = np.asarray(coords, dtype=np.float64)
    if atom_idxs is not None:
        coords = coords[np.asarray(atom_idxs)]
    b = np.diagonal(np.asarray(box, dtype=np.float64))
    inv_b = 1.0 / b
    inv_bin_width = min(inv_b[0], inv_b[1], inv_b[2]) * (HILBERT_GRID_DIM - 1.0)
    x = coords - b * np.floor(coords * inv_b)
    bins = (x * inv_bin_width).astype(np.uint32)  # truncation, as static_cast<unsigned int>
    flat = bins[:, 0].astype(np.int64) * HILBERT_GRID_DIM * HILBERT_GRID_DIM + bins[:, 1].astype(np.int64) * HILBERT_GRID_DIM + bins[:, 2]
    return lut()[flat]


def sort_perm(coords, box, atom_idxs=None):
    """perm = values of a stable sort of (key, atom_idx) pairs, hilbert_sort.cu:69-80."""
    N = np.asarray(coords).shape[0]
    vals = np.arange(N, dtype=np.uint32) if atom_idxs is None else np.asarray(atom_idxs, dtype=np.uint32)
    k = keys(coords, box, atom_idxs)
    order = np.argsort(k, kind="stable")
    return vals[order]
