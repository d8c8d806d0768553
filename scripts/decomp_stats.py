"""CPU costing of tile-kernel decompositions on a DHFR-shaped frame (no GPU needed).

Builds the frame the bench times (testsystems.dhfr_shaped_box, jittered water lattice + solute; atoms displaced by a thermal
random walk to stand in for the steps since the list build), Hilbert-sorts it, builds the per-32-row-block column lists at
cutoff + padding exactly as the library's list kernel defines them (column atom listed iff within the list cutoff of at least
one row atom, upper triangle), and prints for each candidate decomposition what decides its cost:

  lane-owned columns   occupancy = sum popcount / (64 * max popcount) per item (32 rows x 64 consecutive listed columns)
  ... with columns regrouped by hit count inside chunks of C listed columns (C = 128 .. all of the row block)
  16- / 8-row items    slots, items and slot occupancy
  row-uniform rounds   rows with no hit at all in an item (skippable rounds)

usage: python scripts/decomp_stats.py [padding=0.18] [drift_nm=0.03] [row_blocks_sampled=120]
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hilbert  # noqa: E402  (a costing script, not product code)
from timemachine_amd import testsystems as ts  # noqa: E402


def main():
    pad = float(sys.argv[1]) if len(sys.argv) > 1 else 0.18
    drift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
    n_sample = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    s = ts.dhfr_shaped_box()
    rng = np.random.default_rng(11)
    L = np.diagonal(s.box).copy()
    # "equilibrate": the lattice's order is removed by a 0.08 nm rigid-molecule-free jitter (costing only; the GPU-side tools
    # use an MD-equilibrated frame and give the same occupancies to two digits where they overlap: 0.297 at padding 0.18)
    x0 = s.coords + rng.normal(scale=0.05, size=s.coords.shape)
    perm = hilbert.sort_perm(x0, s.box)
    xb = x0[perm]  # positions at list-build time, Hilbert order
    xn = xb + rng.normal(scale=drift / np.sqrt(3.0), size=xb.shape)  # positions "now"
    N = len(xb)
    rc, rl = s.cutoff, s.cutoff + pad
    wrap = lambda a: a - L * np.floor(a / L)
    tree_b = cKDTree(wrap(xb), boxsize=L)
    nb = (N + 31) // 32
    blocks = np.sort(rng.choice(nb, size=min(n_sample, nb), replace=False))

    def mind(a, b):
        d = a[:, None, :] - b[None, :, :]
        d -= L * np.rint(d / L)
        return (d * d).sum(-1)

    tot_pairs = 0
    st = {}  # name -> [useful, issued]
    def acc(name, useful, issued):
        u = st.setdefault(name, [0, 0])
        u[0] += useful
        u[1] += issued

    zero_cols = listed_cols = 0
    pop_hist = np.zeros(33, dtype=np.int64)
    skip_rows = tot_rows = 0
    sub_stats = {16: [0, 0, 0], 8: [0, 0, 0]}  # slots, items, pairs
    for rb in blocks:
        r0, r1 = rb * 32, min(rb * 32 + 32, N)
        rows_b, rows_n = xb[r0:r1], xn[r0:r1]
        cand = np.unique(np.concatenate(tree_b.query_ball_point(wrap(rows_b), rl)))
        cand = cand[cand >= r0]
        d2b = mind(rows_b, xb[cand])
        keep = (d2b < rl * rl).any(0)
        cols = cand[keep]
        d2 = mind(rows_n, xn[cols])  # [rows, cols] now
        hit = d2 < rc * rc
        # upper triangle on the diagonal block: row < col
        ri = np.arange(r0, r1)[:, None]
        hit &= ri < cols[None, :]
        pc = hit.sum(0)  # per column popcount
        listed_cols += len(cols)
        zero_cols += int((pc == 0).sum())
        np.add.at(pop_hist, pc, 1)
        tot_pairs += int(pc.sum())
        n_items = (len(cols) + 63) // 64
        # current design: slots
        acc("slots_32x64", int(pc.sum()), n_items * 2048)
        # lane-owned, list order
        for c0 in range(0, len(cols), 64):
            p = pc[c0:c0 + 64]
            acc("lane_owned_list_order", int(p.sum()), 64 * int(p.max()))
            h = hit[:, c0:c0 + 64]
            skip_rows += int((h.sum(1) == 0).sum())
            tot_rows += h.shape[0]
        # lane-owned with regrouping by popcount inside chunks of C columns (zero-hit columns drop out of the pops)
        for C in (128, 256, 512, 1 << 20):
            for c0 in range(0, len(cols), C):
                p = np.sort(pc[c0:c0 + C])[::-1]
                for v0 in range(0, len(p), 64):
                    q = p[v0:v0 + 64]
                    acc(f"lane_owned_sorted_chunk{C if C < 1 << 20 else 'ALL'}", int(q.sum()), 64 * int(q.max()))
        # class split (stable, few classes) inside a chunk: classes by popcount thresholds
        for C in (256, 512, 1 << 20):
            for edges in ((0, 1, 9, 19, 33), (0, 1, 5, 9, 13, 17, 21, 25, 33)):
                name = f"lane_owned_{len(edges) - 2}classes_chunk{C if C < 1 << 20 else 'ALL'}"
                for c0 in range(0, len(cols), C):
                    p = pc[c0:c0 + C]
                    for lo, hi in zip(edges[1:-1], edges[2:]):
                        q = p[(p >= lo) & (p < hi)]
                        for v0 in range(0, len(q), 64):
                            w = q[v0:v0 + 64]
                            acc(name, int(w.sum()), 64 * int(w.max()))
        # smaller row blocks: lists per sub-block (column listed iff within rl of one of ITS rows at build time)
        for R in (16, 8):
            for q0 in range(0, r1 - r0, R):
                k = (d2b[q0:q0 + R][:, keep] < rl * rl).any(0)
                ncol = int(k.sum())
                it = (ncol + 63) // 64
                sub_stats[R][0] += it * 64 * R
                sub_stats[R][1] += it
                sub_stats[R][2] += int(hit[q0:q0 + R][:, k].sum())
    scale = nb / len(blocks)
    print(f"frame: N {N}, padding {pad}, drift {drift} nm, {len(blocks)} of {nb} row blocks sampled")
    print(f"pairs inside cutoff (scaled to all blocks): {tot_pairs * scale / 1e6:.2f} M; listed columns per row block {listed_cols / len(blocks):.0f}; "
          f"columns with no hit {zero_cols / listed_cols:.3f}")
    print("popcount histogram per listed column (0..32):", pop_hist.tolist())
    print(f"rows with no hit in an item (row-uniform rounds that could be skipped): {skip_rows / tot_rows:.3f}")
    for name, (u, i) in st.items():
        print(f"{name:42s} occupancy {u / i:.3f}   lane-iterations per pair {i / max(u, 1):.2f}")
    for R, (slots, items, pairs) in sub_stats.items():
        print(f"{R:2d}-row items: slots {slots * scale / 1e6:.1f} M, items {items * scale / 1e3:.1f} k, slot occupancy {pairs / slots:.3f}")


if __name__ == "__main__":
    main()
