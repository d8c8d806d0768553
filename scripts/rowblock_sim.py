"""CPU model of the row-block kernel's launch time, for costing unit sizes, piece splits and unit-to-workgroup deals before
building them (no GPU needed).

The frame and the lists are scripts/decomp_stats.py's (the bench's box jittered by a thermal random walk, the list rule of
k_find_ixns, ALL row blocks).  Per unit (row block, <= UCAP listed columns) the model sorts the columns by hit count as the kernel
does, forms virtual items and pieces, deals the pieces to W waves longest-first, and charges the measured constants of
`scripts/rb_timing.py` (EXPERIMENTS.md, round 4): cycles per trip inside the loop, per piece, per filter slice, per unit.  A launch
is the longest workgroup under the chosen deal.  Prints, per candidate: units, mean / max workgroup cycles, the launch in us.

usage: python scripts/rowblock_sim.py [padding=0.18] [drift_nm=0.03]
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hilbert  # noqa: E402  (a costing script, not product code)
from timemachine_amd import testsystems as ts  # noqa: E402

CLOCK_GHZ = 2.4
# measured on the device (f64, 2 x 8 waves, units <= 1024): scripts/rb_timing.py
CYC_TRIP = 1228.0      # per trip inside the pop loop, four waves per SIMD
CYC_PIECE = 1312.0     # record fetch, ticket, column store per piece
CYC_SLICE = 3700.0     # filter: 7.4k per unit of 2 slices per wave (includes waiting for the records)
CYC_SORT = 1400.0
CYC_FLUSH_PER_64 = 3900.0 / 16.0  # flush: 3.9k per unit of 16 chunks on 8 waves -> per 64-column chunk of one wave's share
CYC_DECODE = 2900.0
CYC_PROLOGUE = 5000.0
LAUNCH_US = 2.0        # dispatch + ramp of the launch itself


def unit_masks(pad, drift):
    s = ts.dhfr_shaped_box()
    rng = np.random.default_rng(11)
    L = np.diagonal(s.box).copy()
    x0 = s.coords + rng.normal(scale=0.05, size=s.coords.shape)
    perm = hilbert.sort_perm(x0, s.box)
    xb = x0[perm]
    xn = xb + rng.normal(scale=drift / np.sqrt(3.0), size=xb.shape)
    N = len(xb)
    rc, rl = s.cutoff, s.cutoff + pad
    wrap = lambda a: a - L * np.floor(a / L)
    tree = cKDTree(wrap(xb), boxsize=L)
    nb = (N + 31) // 32
    out = []
    for rb in range(nb):
        r0, r1 = rb * 32, min(rb * 32 + 32, N)
        cand = np.unique(np.concatenate(tree.query_ball_point(wrap(xb[r0:r1]), rl)))
        cand = cand[cand >= r0]
        d = xb[r0:r1][:, None, :] - xb[cand][None, :, :]
        d -= L * np.rint(d / L)
        cols = cand[((d * d).sum(-1) < rl * rl).any(0)]
        d = xn[r0:r1][:, None, :] - xn[cols][None, :, :]
        d -= L * np.rint(d / L)
        hit = (d * d).sum(-1) < rc * rc
        hit &= np.arange(r0, r1)[:, None] < cols[None, :]
        out.append(hit)  # [rows, listed columns] in list order
    return out


def unit_cost(hit, waves, split_pop, quarter_pop=99):
    """cycles of one unit on `waves` waves: (critical path, wave-cycles of work)"""
    n_cols = hit.shape[1]
    pc = hit.sum(0)
    order = np.argsort(-pc, kind="stable")
    pcs = pc[order]
    nnz = int((pcs > 0).sum())
    pieces = []  # trips per piece
    for v0 in range(0, nnz, 64):
        cols = order[v0:v0 + 64]
        top = int(pcs[v0])
        if top > quarter_pop:
            parts = 4
        elif top > split_pop:
            parts = 2
        else:
            parts = 1
        rows_per = hit.shape[0] // parts if parts > 1 else hit.shape[0]
        for p in range(parts):
            sub = hit[p * rows_per:(p + 1) * rows_per if parts > 1 else None][:, cols]
            pieces.append(int(sub.sum(0).max()) if sub.size else 0)
    pieces.sort(reverse=True)
    load = np.zeros(waves)
    for t in pieces:  # the waves draw pieces heaviest first
        w = int(load.argmin())
        load[w] += t * CYC_TRIP + CYC_PIECE
    pops = load.max() if len(pieces) else 0.0
    slices = (n_cols + 63) // 64
    filt = np.ceil(slices / waves) * CYC_SLICE
    flush = np.ceil(slices / waves) * CYC_FLUSH_PER_64 * (16 / 8) * (8 / waves) if slices else 0.0
    trips = sum(pieces)
    return CYC_DECODE + filt + CYC_SORT + pops + flush, trips, int(pc.sum())


def simulate(masks, ucap, waves, wgs, split_pop, deal, quarter_pop=99):
    units = []  # (critical cycles, trips, pairs, columns)
    for hit in masks:
        count = hit.shape[1]
        if count == 0:
            continue
        parts = (count + ucap - 1) // ucap
        usize = (((count + parts - 1) // parts) + 63) & ~63
        for p in range(parts):
            sub = hit[:, p * usize:(p + 1) * usize]
            if sub.shape[1]:
                c, t, pr = unit_cost(sub, waves, split_pop, quarter_pop)
                units.append((c, t, pr, sub.shape[1]))
    cost = np.array([u[0] for u in units])
    cols = np.array([u[3] for u in units])
    load = np.zeros(wgs)
    if deal == "serpentine":  # the kernel's: position k of workgroup b = unit k G + (k even ? b : G - 1 - b), row-block order
        for u, c in enumerate(cost):
            k, r = divmod(u, wgs)
            load[r if k % 2 == 0 else wgs - 1 - r] += c
    elif deal == "by_columns":  # units sorted by column count (known before the launch), dealt serpentine
        for u, i in enumerate(np.argsort(-cols, kind="stable")):
            k, r = divmod(u, wgs)
            load[r if k % 2 == 0 else wgs - 1 - r] += cost[i]
    elif deal == "lpt_columns":  # longest-processing-time on column counts as the cost estimate
        est = np.zeros(wgs)
        for i in np.argsort(-cols, kind="stable"):
            w = int(est.argmin())
            est[w] += cols[i]
            load[w] += cost[i]
    elif deal == "dynamic":  # a device-wide ticket in row-block order, drawn when a workgroup runs dry (no latency charged)
        for c in cost:
            w = int(load.argmin())
            load[w] += c
    elif deal == "dynamic_sorted":  # the same over units sorted by column count
        for i in np.argsort(-cols, kind="stable"):
            w = int(load.argmin())
            load[w] += cost[i]
    load += CYC_PROLOGUE
    trips = sum(u[1] for u in units)
    pairs = sum(u[2] for u in units)
    return len(units), load.mean(), load.max(), trips, pairs


def main():
    pad = float(sys.argv[1]) if len(sys.argv) > 1 else 0.18
    drift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
    masks = unit_masks(pad, drift)
    print(f"frame: {sum(m.shape[1] for m in masks)} listed columns, {sum(int(m.sum()) for m in masks) / 1e6:.2f} M pairs")
    print("ucap waves wgs split quarter deal            units  mean_kcyc  max_kcyc  launch_us  trip_occupancy")
    for ucap, waves, wgs, split_pop, quarter in (
        (1024, 8, 512, 16, 99), (1024, 8, 512, 8, 99), (1024, 8, 512, 8, 20), (768, 8, 512, 8, 99), (512, 8, 512, 8, 99), (512, 4, 1024, 8, 99),
        (1536, 8, 512, 8, 99), (2048, 16, 256, 8, 99), (1024, 16, 256, 8, 99), (1024, 16, 256, 8, 20),
    ):
        for deal in ("serpentine", "by_columns", "lpt_columns", "dynamic", "dynamic_sorted"):
            n, mean, mx, trips, pairs = simulate(masks, ucap, waves, wgs, split_pop, deal, quarter)
            print(f"{ucap:5d} {waves:4d} {wgs:5d} {split_pop:4d} {quarter:6d}  {deal:15s} {n:5d}  {mean / 1e3:8.1f}  {mx / 1e3:8.1f}  {mx / CLOCK_GHZ / 1e3 + LAUNCH_US:8.1f}   {pairs / (64.0 * trips):.3f}")


if __name__ == "__main__":
    main()
