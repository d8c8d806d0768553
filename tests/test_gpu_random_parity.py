"""GPU test (``-m gpu``): randomly drawn small systems against the CPU oracle -- sizes that are no multiple of anything, orthorhombic
boxes down to less than twice the cutoff, atoms left several box lengths outside the home box, clashing pairs (below the f64 force
table), sites without Lennard-Jones or charge, 4D offsets, random exclusions with random scales, random atom subsets and interaction
groups, a random cutoff.  The golden vectors (tests/golden/) pin a handful of hand-made systems to the reference's own Python; this
covers the space between them, against oracle/ref_potentials.py (itself pinned to the reference by those goldens).
Tolerances: the reference's (tests/common.py:275-334 as used in tests/nonbonded/test_nonbonded.py): 1e-8 relative in f64, 1e-4 in
f32 on forces measured against the per-atom force norm.  scripts/fuzz_parity.py runs the same over many more seeds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


def draw_system(seed, gentle=False):
    """a random periodic system: jittered lattice (no pair closer than ~0.07 nm unless it is made an exclusion), water-like density.
    ``gentle`` (the f32 cases): no clashes and every atom inside the home box.  An f32 pair function sees its distance with an error of
    a few ulp of the COORDINATES (2e-7 nm at |x| = 2 nm), which a clashing pair's r^-13 force multiplies by 13 / r: comparing f32
    against an f64 oracle on clashing or far-out-of-box atoms measures the conditioning of the input, not the kernel (the reference's
    f32 kernels form the same f32 differences, k_nonbonded.cuh:168-180); the f64 cases keep all of it."""
    rng = np.random.default_rng(seed)
    N = int(rng.choice([3, 17, 31, 32, 33, 64, 97, 130, 257, 411, 640]))
    cutoff = float(rng.choice([0.7, 0.9, 1.0, 1.2, 1.3]))
    n_side = int(np.ceil(N ** (1 / 3)))
    spacing = rng.uniform(0.27, 0.32) if gentle else rng.uniform(0.2, 0.3)
    aspect = rng.uniform(0.8, 1.25, size=3)
    grid = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)[rng.permutation(n_side ** 3)[:N]]
    x = (grid + 0.5) * spacing * aspect + rng.normal(0.0, 0.015 if gentle else 0.04, (N, 3))
    box = np.diag(n_side * spacing * aspect)
    if not gentle:
        x += rng.integers(-3, 4, size=(N, 3)) * np.diagonal(box) * (rng.random((N, 1)) < 0.2)  # some atoms far outside the home box
    params = np.zeros((N, 4))
    params[:, 0] = rng.normal(0.0, 0.5, N) * (rng.random(N) > 0.15)
    params[:, 1] = rng.uniform(0.05, 0.11 if gentle else 0.17, N)
    params[:, 2] = rng.uniform(0.1, 1.0, N) * (rng.random(N) > 0.3)
    params[:, 3] = rng.choice([0.0, 0.0, 0.25, 0.5], N) * cutoff * rng.random(N)
    # clashes stay inside the fixed-point accumulators' range: a force component beyond 2^27 kJ/mol/nm wraps in the 2^36-scaled
    # u64 sums (k_fixed_point.cuh:10-24 -- the reference's as well), so a bare pair force above 2^25 -- an interaction group's clash,
    # or a partly excluded pair whose all-pairs part alone leaves the range -- compares the oracle with an overflow, not with the
    # kernel (scripts/fuzz_parity.py, seeds 20148 / 21158).  Such atoms get smaller sigmas until no pair is beyond 2^25.
    for _ in range(40):
        g_lj, g_es = pair_force_matrix(x, params, box, cutoff, 2.0)
        big = np.argwhere(g_lj + g_es > 2.0 ** 25)
        if len(big) == 0:
            break
        params[np.unique(big), 1] *= 0.85
    # exclusions: the closest pairs (so that whatever clashes is at least partly excluded, like bonded neighbours) + random ones
    L = np.diagonal(box)
    d = x[:, None, :] - x[None, :, :]
    d -= L * np.rint(d / L)
    r = np.sqrt((d ** 2).sum(-1)) + np.eye(N) * 10.0
    close = np.argwhere(np.triu(r < 0.16))
    extra = rng.integers(0, N, size=(N // 2, 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    ex = np.unique(np.sort(np.concatenate([close, extra]), axis=1), axis=0).astype(np.int32)
    scales = rng.choice([0.0, 0.5, 1.0 / 1.2, 1.0], size=(len(ex), 2))
    scales[r[ex[:, 0], ex[:, 1]] < 0.1] = 1.0  # overlapping atoms: fully excluded, as bonded atoms are
    return dict(N=N, x=x, box=box, params=params, cutoff=cutoff, beta=2.0, ex=ex, scales=scales, rng=rng)


def pair_force_matrix(x, prm, box, cutoff, beta):
    """|dU_ij/dd_ij| of every pair (no exclusions applied), Lennard-Jones and electrostatic part"""
    import torch

    from oracle import ref_potentials as rp

    N = len(x)
    xt, pt = torch.as_tensor(x), torch.as_tensor(prm)
    d3 = rp.delta_r(xt[:, None, :], xt[None, :, :], torch.as_tensor(np.diagonal(box).copy()))
    dw = pt[:, 3][:, None] - pt[None, :, 3]
    d = torch.sqrt((d3 * d3).sum(-1) + dw * dw + torch.eye(N) * 100.0).requires_grad_(True)
    lj, es = rp._pair_energies(d, pt[:, 0][:, None] * pt[None, :, 0], pt[:, 1][:, None] + pt[None, :, 1], pt[:, 2][:, None] * pt[None, :, 2], beta, cutoff)
    g_lj = torch.autograd.grad(lj.sum(), d, retain_graph=True)[0].abs().numpy()
    g_es = torch.autograd.grad(es.sum(), d)[0].abs().numpy()
    return g_lj, g_es


def pair_force_sums(x, prm, box, cutoff, beta, ex=None, scales=None, rows=None, cols=None):
    """per atom, the sum over its pairs of |dU_ij/dd_ij| (each pair's force magnitude, exclusion scales applied): what an f32 pair
    function's rounding is proportional to -- an atom squeezed between two clashing neighbours has a small NET force made of two
    large ones, and its f32 error is a few ulp of THOSE"""
    N = len(x)
    g_lj, g_es = pair_force_matrix(x, prm, box, cutoff, beta)
    if ex is not None and len(ex):
        for (i, j), (sq, sl) in zip(ex, scales):
            g_es[i, j] *= abs(1.0 - sq); g_es[j, i] *= abs(1.0 - sq)
            g_lj[i, j] *= abs(1.0 - sl); g_lj[j, i] *= abs(1.0 - sl)
    g = g_lj + g_es
    if rows is not None:
        m = np.zeros((N, N), dtype=bool)
        m[np.ix_(rows, cols)] = True
        g = g * (m | m.T)
    return g.sum(1)


def cutoff_off_straddling_pairs(x, prm, box, cutoff):
    """the cutoff, moved up in steps of 3e-4 nm until no pair's d^2 (4D, minimum image) lies within 2e-6 nm^2 of cutoff^2 as an f32
    kernel reads it"""
    L = np.diagonal(box)
    d = x[:, None, :] - x[None, :, :]
    d -= L * np.rint(d / L)
    d2 = (d ** 2).sum(-1) + (prm[:, 3][:, None] - prm[None, :, 3]) ** 2
    while np.any(np.abs(d2 - np.float64(np.float32(cutoff)) ** 2) < 2e-6):
        cutoff += 3e-4
    return cutoff


def check(tag, got, ref, precision, pair_sums):
    du_dx, du_dp, u = got
    ref_u, ref_dx, ref_dp = ref
    f64 = precision == np.float64
    tol = 1e-8 if f64 else 1e-4
    assert abs(u - ref_u) <= tol * max(1.0, abs(ref_u)) * (1 if f64 else 5), (tag, u, ref_u)
    nrm = np.maximum(np.linalg.norm(ref_dx, axis=1, keepdims=True), 1.0)
    if not f64:  # + 2e-6 of the pair forces the net force is made of (a few f32 ulp each; the sum itself is exact: integers)
        nrm = nrm + 2e-2 * pair_sums[:, None]
    err = (np.abs(du_dx - ref_dx) / nrm).max()
    assert err <= tol, (tag, "du_dx", err)
    perr = (np.abs(du_dp - ref_dp) / np.maximum(np.abs(ref_dp), 1.0)).max()
    assert perr <= (1e-7 if f64 else 2e-3), (tag, "du_dp", perr)


def run_case(seed, precision):
    from oracle import ref_potentials as rp
    from timemachine_amd import potentials as P

    s = draw_system(seed, gentle=precision == np.float32)
    N, x, box, prm, cutoff, beta, rng = s["N"], s["x"], s["box"], s["params"], s["cutoff"], s["beta"], s["rng"]
    if precision == np.float32:
        x = x.astype(np.float32).astype(np.float64)
        prm = prm.astype(np.float32).astype(np.float64)
        # the cutoff is a step in the force (the electrostatic part is not switched off at it): a pair whose d^2 lies within f32
        # rounding of cutoff^2 is inside for one precision and outside for the other, and the comparison measures that pair's whole
        # force (scripts/fuzz_parity.py, seed 21355: d^2 - cutoff^2 = -8.8e-8 in f64, +1.8e-7 in f32).  Move the cutoff off such pairs.
        cutoff = cutoff_off_straddling_pairs(x, prm, box, cutoff)
    tag = f"seed {seed} N {N} cutoff {cutoff} box {np.diagonal(box).round(2)}"
    nb = P.Nonbonded(N, s["ex"], s["scales"], beta, cutoff).to_gpu(precision).unbound_impl
    sums = pair_force_sums(x, prm, box, cutoff, beta, s["ex"], s["scales"])
    check(tag + " nonbonded", nb.execute(x, prm, box), rp.nonbonded(x, prm, box, s["ex"], s["scales"], beta, cutoff), precision, sums)
    if N >= 17:
        sub = np.sort(rng.choice(N, size=int(rng.integers(2, N)), replace=False)).astype(np.int32)
        nbs = P.Nonbonded(N, s["ex"], s["scales"], beta, cutoff, atom_idxs=sub).to_gpu(precision).unbound_impl
        check(tag + " subset", nbs.execute(x, prm, box), rp.nonbonded(x, prm, box, s["ex"], s["scales"], beta, cutoff, atom_idxs=sub), precision, sums)
        rows = np.sort(rng.choice(N, size=int(rng.integers(1, N // 2)), replace=False)).astype(np.int32)
        cols = np.setdiff1d(np.arange(N, dtype=np.int32), rows)
        if rng.random() < 0.5:
            cols = np.sort(rng.choice(cols, size=max(1, len(cols) // 2), replace=False)).astype(np.int32)
        # an interaction group has no exclusions: pairs of overlapping atoms across the groups would be bare clashes
        L = np.diagonal(box)
        d = x[rows][:, None, :] - x[cols][None, :, :]
        d -= L * np.rint(d / L)
        if np.sqrt((d ** 2).sum(-1)).min() > 0.09:
            ig = P.NonbondedInteractionGroup(N, rows, beta, cutoff, col_atom_idxs=cols).to_gpu(precision).unbound_impl
            check(tag + " group", ig.execute(x, prm, box), rp.nonbonded_interaction_group(x, prm, box, rows, beta, cutoff, col_atom_idxs=cols), precision,
                  pair_force_sums(x, prm, box, cutoff, beta, rows=rows, cols=cols))


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_systems_against_the_oracle(co, seed, precision):
    run_case(seed, precision)


def run_bonded_case(seed, precision):
    """random index lists and parameters of the per-term potentials (repeated atoms across terms, terms in random order, zero force
    constants, r0 = 0 bonds, torsions of period 1..6) on a jittered lattice, against the oracle"""
    from oracle import ref_potentials as rp
    from timemachine_amd import potentials as P

    s = draw_system(seed, gentle=True)
    N, box, rng = s["N"], s["box"], s["rng"]
    if N < 17:
        return
    x = s["x"].astype(np.float32).astype(np.float64) if precision == np.float32 else s["x"]
    f64 = precision == np.float64
    # neighbours on the lattice make physical-looking terms: chains of nearby atoms
    L = np.diagonal(box)
    d = x[:, None, :] - x[None, :, :]
    r = np.sqrt((d ** 2).sum(-1)) + np.eye(N) * 10.0  # (bonded terms do not see the box: plain distances, bonded.py:34-79)
    near = np.argsort(r, axis=1)[:, :3]
    M = int(rng.integers(1, 3 * N))
    a = rng.integers(0, N, M)
    b = near[a, rng.integers(0, 3, M)]
    c = near[b, rng.integers(0, 3, M)]
    e = near[c, rng.integers(0, 3, M)]

    def compare(tag, pot, prm, ref, rtol, ptol, dp_mask=None, extra=None):
        du_dx, du_dp, u = pot.to_gpu(precision).unbound_impl.execute(x, prm, box)
        ref_u, ref_dx, ref_dp = ref
        assert abs(u - ref_u) <= (1e-8 if f64 else 2e-5) * max(1.0, abs(ref_u)), (seed, tag, u, ref_u)
        nrm = np.maximum(np.linalg.norm(ref_dx, axis=1, keepdims=True), 1.0)
        if extra is not None:  # a per-atom absolute allowance on top of rtol * nrm (an input-rounding model, see the caller)
            nrm = nrm + extra[:, None] / rtol
        assert (np.abs(du_dx - ref_dx) / nrm).max() <= rtol, (seed, tag, "du_dx", (np.abs(du_dx - ref_dx) / nrm).max())
        perr = np.abs(np.asarray(du_dp) - ref_dp) / np.maximum(np.abs(ref_dp), 1.0)
        if dp_mask is not None:
            perr = perr * dp_mask
        assert perr.max() <= ptol, (seed, tag, "du_dp", perr.max())

    rtol, ptol = (1e-7, 1e-7) if f64 else (1e-4, 2e-3)

    def r32(p):
        """f32 cases: parameters as an f32 kernel reads them, for the oracle too (as the nonbonded cases do): rounding an input is not
        the kernel's arithmetic"""
        return p if f64 else p.astype(np.float32).astype(np.float64)

    bonds = np.stack([a, b], 1).astype(np.int32)
    # (f32: softer force constants -- the absolute error of an f32 bond force is k * eps_f32 * d whatever the arithmetic, and the
    # tolerance is relative to a force norm with a floor of 1)
    kb_max, ka_max = (5e4, 500.0) if f64 else (1e3, 60.0)
    bp = r32(np.stack([rng.uniform(0.0, kb_max, M) * (rng.random(M) > 0.1), rng.uniform(0.08, 0.3, M) * (rng.random(M) > 0.1)], 1))
    # (du/dr0 of a bond with r0 == 0: the reference's kernel reports -k d, k_harmonic_bond.cuh:52, its Python -- jnp.where picks the
    # r0-free branch, bonded.py:44 -- reports 0; this build follows the kernel, the oracle the Python: not compared)
    compare("bond", P.HarmonicBond(bonds), bp, rp.harmonic_bond(x, bp, box, bonds), rtol, ptol, dp_mask=np.stack([np.ones(M), bp[:, 1] != 0], 1))
    ok = (a != c)
    angles = np.stack([a, b, c], 1)[ok].astype(np.int32)
    if len(angles):
        ap = r32(np.stack([rng.uniform(0.0, ka_max, len(angles)), rng.uniform(1.0, 3.0, len(angles)), rng.choice([0.0, 1e-3], len(angles))], 1))
        compare("angle", P.HarmonicAngle(angles), ap, rp.harmonic_angle(x, ap, box, angles), rtol if f64 else 4e-4, ptol)  # (f32: short arms, angles near pi)
    ok = (a != c) & (a != e) & (b != e)
    tors = np.stack([a, b, c, e], 1)[ok].astype(np.int32)
    if len(tors):
        # (near-collinear triples make the dihedral ill-conditioned for any implementation: skip them)
        rij, rkj, rkl = x[tors[:, 1]] - x[tors[:, 0]], x[tors[:, 1]] - x[tors[:, 2]], x[tors[:, 3]] - x[tors[:, 2]]
        s1 = np.linalg.norm(np.cross(rij, rkj), axis=1) / (np.linalg.norm(rij, axis=1) * np.linalg.norm(rkj, axis=1))
        s2 = np.linalg.norm(np.cross(rkj, rkl), axis=1) / (np.linalg.norm(rkj, axis=1) * np.linalg.norm(rkl, axis=1))
        tors = tors[(s1 > 0.2) & (s2 > 0.2)]
    if len(tors):
        tp = r32(np.stack([rng.uniform(0.0, 20.0, len(tors)), rng.uniform(-np.pi, np.pi, len(tors)), rng.integers(1, 7, len(tors)).astype(np.float64)], 1))
        extra = None
        if not f64:
            # f32: the kernel (the reference's, k_periodic_torsion.cuh:49-131) forms arg = n * angle - phase in f32: |arg| <= (n + 1) pi, so
            # arg carries (n + 1) pi 2^-23, and the force k n sin(arg) d(angle)/dx that much times k n |d(angle)/dx| -- stiff,
            # high-period torsions with a short lever arm reach 3-7e-4 of a small net force (scripts/fuzz_parity.py: ~2 of 1 000
            # seeds beyond a flat 3e-4; the reference kernel's formula evaluated in numpy f32 reproduces the GPU's error on those
            # atoms digit for digit).  The bar per atom: 1e-4 of max(|F|, 1) + twice that rounding model summed over its terms.
            ti, tj, tk, tl = tors.T
            rij, rkj, rkl = x[tj] - x[ti], x[tj] - x[tk], x[tl] - x[tk]
            n1, n2 = np.cross(rij, rkj), np.cross(rkj, rkl)
            q = (rkj * rkj).sum(1)
            dR0 = (np.sqrt(q) / (n1 * n1).sum(1))[:, None] * n1
            dR3 = (-np.sqrt(q) / (n2 * n2).sum(1))[:, None] * n2
            aa, bb = (rij * rkj).sum(1) / q, (rkl * rkj).sum(1) / q
            dR1 = (aa - 1)[:, None] * dR0 - dR3 * bb[:, None]
            dR2 = (bb - 1)[:, None] * dR3 - dR0 * aa[:, None]
            w = tp[:, 0] * tp[:, 2] * (tp[:, 2] + 1) * 2.0 * np.pi * 2.0 ** -23
            extra = np.zeros(N)
            for idx, dR in ((ti, dR0), (tj, dR1), (tk, dR2), (tl, dR3)):
                np.add.at(extra, idx, w * np.linalg.norm(dR, axis=1))
        compare("torsion", P.PeriodicTorsion(tors), tp, rp.periodic_torsion(x, tp, box, tors), rtol, ptol, extra=extra)
        cp = r32(rng.uniform(0.0, 100.0, len(tors)))
        compare("chiral atom", P.ChiralAtomRestraint(tors), cp, rp.chiral_atom_restraint(x, cp, box, tors), rtol * 10, ptol)
    # precomputed pair list: per-pair (q_ij, sig_ij, eps_ij, w_ij), periodic, cutoff
    pairs = np.stack([a, near[a, 2]], 1).astype(np.int32)
    pp = r32(np.stack([rng.normal(0.0, 0.3, M), rng.uniform(0.1, 0.22, M), rng.uniform(0.0, 1.0, M) * (rng.random(M) > 0.3), rng.choice([0.0, 0.1, 0.4], M)], 1))
    compare("precomputed", P.NonbondedPairListPrecomputed(pairs, s["beta"], s["cutoff"]), pp,
            rp.nonbonded_pair_list_precomputed(x, pp, box, pairs, s["beta"], s["cutoff"]), rtol if f64 else 5e-4, ptol)
    r_min = rng.uniform(0.0, 0.3, M)
    fb = r32(np.stack([rng.uniform(0.0, 1e3, M), r_min, r_min + rng.uniform(0.0, 0.2, M)], 1))
    compare("flat bottom", P.FlatBottomBond(bonds), fb, rp.flat_bottom_bond(x, fb, box, bonds), rtol, ptol)


@pytest.mark.parametrize("precision", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_random_bonded_terms_against_the_oracle(co, seed, precision):
    run_bonded_case(seed, precision)
