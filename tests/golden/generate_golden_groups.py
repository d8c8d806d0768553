"""Generates tests/golden/groups.npz: interaction groups, precomputed pair lists and chiral restraints, by running the
REFERENCE's own Python implementation (same protocol as generate_golden.py, which documents what comes from where).

    python tests/golden/generate_golden_groups.py        (build container only: needs /root/reference)

Energies: timemachine/potentials/nonbonded.py (nonbonded_interaction_groups, nonbonded_on_precomputed_pairs) and
chiral_restraints.py (chiral_atom_restraint, chiral_bond_restraint), unmodified, under the numpy `jax` shim.
Gradients: oracle/ref_potentials.py (torch autograd), asserted against the reference energy (1e-12 relative) and against
central finite differences of the REFERENCE energy before anything is written.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _jax_numpy_shim  # noqa: E402

_tmp = tempfile.mkdtemp(prefix="jaxshim_")
_jax_numpy_shim.materialise(_tmp)
sys.path[:0] = [_tmp, REF]

from timemachine.potentials import bonded as ref_bonded  # noqa: E402
from timemachine.potentials import chiral_restraints as ref_chiral  # noqa: E402
from timemachine.potentials import nonbonded as ref_nonbonded  # noqa: E402

from generate_golden import random_nb_system, rel  # noqa: E402
from oracle import ref_potentials as rp  # noqa: E402

np.seterr(all="ignore")


def check_fd(name, f_ref, x, g_oracle, rng, n=12, h=1e-5, tol=2e-6):
    """central finite differences of the REFERENCE energy vs the oracle gradient on n random entries.  Entries that are
    exactly zero are skipped: eps == 0 / q_ij == 0 switch a term off through `where`, where jax.grad (and the kernels)
    report the derivative of the masked branch (0) while a finite difference steps across the kink."""
    flat = x.reshape(-1)
    worst = 0.0
    candidates = np.flatnonzero(flat != 0)
    for k in rng.choice(candidates, size=min(n, candidates.size), replace=False):
        xp, xm = flat.copy(), flat.copy()
        xp[k] += h
        xm[k] -= h
        fd = (f_ref(xp.reshape(x.shape)) - f_ref(xm.reshape(x.shape))) / (2 * h)
        g = g_oracle.reshape(-1)[k]
        err = abs(fd - g) / max(1.0, abs(g))
        worst = max(worst, err)
        assert err < tol, (name, k, fd, g, err)
    print(f"  fd-check {name}: worst rel err {worst:.2e}")


def main():
    rng = np.random.default_rng(20260928)
    beta, cutoff = 2.0, 1.2
    d = {}

    # ---- interaction groups: a 40-atom "ligand" (w != 0 on part of it) against the rest, and against an explicit subset
    n, L = 200, 3.3
    x, p, _, _ = random_nb_system(rng, n, L, cutoff, "zero")
    box = np.eye(3) * L
    rows = np.sort(rng.choice(n, 40, replace=False)).astype(np.int32)
    p[rows[:20], 3] = 0.25 * cutoff
    p[rows[20:30], 3] = cutoff  # fully decoupled in the 4th dimension: d >= cutoff to every w = 0 atom
    cols_sub = np.sort(rng.choice(np.setdiff1d(np.arange(n), rows), 90, replace=False)).astype(np.int32)
    for tag, cols in (("ig_all", None), ("ig_sub", cols_sub)):
        vdw, es = ref_nonbonded.nonbonded_interaction_groups(x, p, box, rows, cols, beta, cutoff)
        u_ref = float(np.sum(vdw) + np.sum(es))
        u, gx, gp = rp.nonbonded_interaction_group(x, p, box, rows, beta, cutoff, cols)
        assert rel(u, u_ref) < 1e-12, (tag, u, u_ref)

        def f_x(xx, cols=cols):
            a, b = ref_nonbonded.nonbonded_interaction_groups(xx, p, box, rows, cols, beta, cutoff)
            return float(np.sum(a) + np.sum(b))

        def f_p(pp, cols=cols):
            a, b = ref_nonbonded.nonbonded_interaction_groups(x, pp, box, rows, cols, beta, cutoff)
            return float(np.sum(a) + np.sum(b))

        check_fd(tag + " du_dx", f_x, x, gx, rng)
        check_fd(tag + " du_dp", f_p, p, gp, rng, h=1e-6, tol=2e-5)
        d.update({f"{tag}_u": u_ref, f"{tag}_du_dx": gx, f"{tag}_du_dp": gp})
        print(f"  {tag}: u={u_ref:.6f}")
    d.update(ig_x=x, ig_params=p, ig_box=box, ig_rows=rows, ig_cols_sub=cols_sub, beta=beta, cutoff=cutoff)

    # ---- precomputed pair list
    B = 300
    pairs = np.stack([rng.permutation(n)[:2] for _ in range(B)]).astype(np.int32)
    pp = np.stack(
        [
            (rng.uniform(size=B) - 0.5) * 138.935456 * 0.25,  # q_ij
            rng.uniform(0.1, 0.4, B),  # sig_ij
            rng.uniform(0.0, 1.0, B),  # eps_ij
            rng.uniform(-0.3, 0.3, B),  # w offsets
        ],
        1,
    )
    pp[:30, 0] = 0.0  # LJ-only pairs
    pp[30:60, 2] = 0.0  # electrostatics-only pairs (the reference's CUDA kernel drops these; its JAX definition does not)
    pp[60:70, 3] = 0.0
    pp = pp.astype(np.float32).astype(np.float64)
    vdw, es = ref_nonbonded.nonbonded_on_precomputed_pairs(x, pp, box, pairs, beta, cutoff)
    u_ref = float(np.sum(vdw) + np.sum(es))
    u, gx, gp = rp.nonbonded_pair_list_precomputed(x, pp, box, pairs, beta, cutoff)
    assert rel(u, u_ref) < 1e-12, (u, u_ref)

    def f_pre(xx, q=pp):
        a, b = ref_nonbonded.nonbonded_on_precomputed_pairs(xx, q, box, pairs, beta, cutoff)
        return float(np.sum(a) + np.sum(b))

    check_fd("precomputed du_dx", f_pre, x, gx, rng)
    check_fd("precomputed du_dp", lambda q: f_pre(x, q), pp, gp, rng, h=1e-6, tol=2e-5)
    d.update(pre_idxs=pairs, pre_params=pp, pre_u=u_ref, pre_du_dx=gx, pre_du_dp=gp)
    print(f"  precomputed: u={u_ref:.6f}")

    # ---- chiral restraints (positions on the 0.1 nm scale so that volumes of both signs occur)
    m = 64
    xc = rng.uniform(0, 1.5, (m, 3)).astype(np.float32).astype(np.float64)
    R = 80
    a_idxs = np.stack([rng.permutation(m)[:4] for _ in range(R)]).astype(np.int32)
    a_k = rng.uniform(10.0, 1000.0, R)
    a_k[:5] = 0.0
    u_ref = float(ref_chiral.chiral_atom_restraint(xc, a_k, None, a_idxs))
    u, gx, gp = rp.chiral_atom_restraint(xc, a_k, None, a_idxs)
    assert rel(u, u_ref) < 1e-12, (u, u_ref)
    check_fd("chiral atom du_dx", lambda xx: float(ref_chiral.chiral_atom_restraint(xx, a_k, None, a_idxs)), xc, gx, rng, h=1e-6, tol=5e-6)
    check_fd("chiral atom du_dp", lambda kk: float(ref_chiral.chiral_atom_restraint(xc, kk, None, a_idxs)), a_k, gp, rng, h=1e-4, tol=5e-6)
    d.update(chiral_x=xc, chiral_atom_idxs=a_idxs, chiral_atom_params=a_k, chiral_atom_u=u_ref, chiral_atom_du_dx=gx, chiral_atom_du_dp=gp)
    print(f"  chiral atom: u={u_ref:.6f} ({int((gp > 0).sum())} of {R} active)")

    b_idxs = np.stack([rng.permutation(m)[:4] for _ in range(R)]).astype(np.int32)
    b_signs = rng.choice([-1, 1], R).astype(np.int32)
    b_k = rng.uniform(10.0, 1000.0, R)
    u_ref = float(ref_chiral.chiral_bond_restraint(xc, b_k, None, b_idxs, b_signs))
    u, gx, gp = rp.chiral_bond_restraint(xc, b_k, None, b_idxs, b_signs)
    assert rel(u, u_ref) < 1e-12, (u, u_ref)
    check_fd("chiral bond du_dx", lambda xx: float(ref_chiral.chiral_bond_restraint(xx, b_k, None, b_idxs, b_signs)), xc, gx, rng, h=1e-6, tol=5e-6)
    d.update(chiral_bond_idxs=b_idxs, chiral_bond_signs=b_signs, chiral_bond_params=b_k, chiral_bond_u=u_ref, chiral_bond_du_dx=gx, chiral_bond_du_dp=gp)
    print(f"  chiral bond: u={u_ref:.6f} ({int((gp > 0).sum())} of {R} active)")

    # ---- flat-bottom / log flat-bottom bonds (PBC) and the centroid restraint
    Bf = 60
    fb_idxs = np.stack([rng.permutation(m)[:2] for _ in range(Bf)]).astype(np.int32)
    fb_box = np.eye(3) * 1.2  # smaller than the coordinate spread: minimum-image distances matter
    fb_p = np.stack([rng.uniform(50, 500, Bf), rng.uniform(0.2, 0.35, Bf), rng.uniform(0.4, 0.6, Bf)], 1)
    u_ref = float(ref_bonded.flat_bottom_bond(xc, fb_p, fb_box, fb_idxs))
    u, gx, gp = rp.flat_bottom_bond(xc, fb_p, fb_box, fb_idxs)
    assert rel(u, u_ref) < 1e-12, (u, u_ref)
    check_fd("flat bottom du_dx", lambda xx: float(ref_bonded.flat_bottom_bond(xx, fb_p, fb_box, fb_idxs)), xc, gx, rng, h=1e-6, tol=5e-6)
    check_fd("flat bottom du_dp", lambda pp: float(ref_bonded.flat_bottom_bond(xc, pp, fb_box, fb_idxs)), fb_p, gp, rng, h=1e-6, tol=5e-6)
    d.update(fb_idxs=fb_idxs, fb_box=fb_box, fb_params=fb_p, fb_u=u_ref, fb_du_dx=gx, fb_du_dp=gp)
    print(f"  flat bottom: u={u_ref:.6f}")
    lfb_beta = 0.4
    # only bonds outside the flat region have a finite log energy (-log(1 - exp(0)) = inf inside): keep those
    nrgs = np.asarray(ref_bonded._flat_bottom_bond_impl(xc, fb_p, fb_box, fb_idxs))
    keep = nrgs > 1e-3
    lfb_idxs, lfb_p = fb_idxs[keep], fb_p[keep]
    u_ref = float(ref_bonded.log_flat_bottom_bond(xc, lfb_p, fb_box, lfb_idxs, lfb_beta))
    u, gx, gp = rp.log_flat_bottom_bond(xc, lfb_p, fb_box, lfb_idxs, lfb_beta)
    assert rel(u, u_ref) < 1e-12, (u, u_ref)
    check_fd("log flat bottom du_dx", lambda xx: float(ref_bonded.log_flat_bottom_bond(xx, lfb_p, fb_box, lfb_idxs, lfb_beta)), xc, gx, rng, h=1e-7, tol=2e-5)
    d.update(lfb_idxs=lfb_idxs, lfb_params=lfb_p, lfb_beta=lfb_beta, lfb_u=u_ref, lfb_du_dx=gx, lfb_du_dp=gp)
    print(f"  log flat bottom: u={u_ref:.6f} ({int(keep.sum())} bonds)")
    ga = rng.choice(m, 9, replace=False).astype(np.int32)
    gb = rng.choice(np.setdiff1d(np.arange(m), ga), 14, replace=False).astype(np.int32)
    for tag, kb, b0 in (("cr", 120.0, 0.25), ("cr0", 75.0, 0.0)):
        u_ref = float(ref_bonded.centroid_restraint(xc, None, None, ga, gb, kb, b0))
        u, gx, _ = rp.centroid_restraint(xc, None, None, ga, gb, kb, b0)
        assert rel(u, u_ref) < 1e-12, (u, u_ref)
        check_fd(tag + " du_dx", lambda xx, kb=kb, b0=b0: float(ref_bonded.centroid_restraint(xx, None, None, ga, gb, kb, b0)), xc, gx, rng, h=1e-6, tol=5e-6)
        d.update({f"{tag}_kb": kb, f"{tag}_b0": b0, f"{tag}_u": u_ref, f"{tag}_du_dx": gx})
        print(f"  centroid restraint b0={b0}: u={u_ref:.6f}")
    d.update(cr_a=ga, cr_b=gb)

    np.savez_compressed(os.path.join(os.environ.get("TM_GOLDEN_OUT", HERE), "groups.npz"), **d)
    shutil.rmtree(_tmp, ignore_errors=True)
    print("done")


if __name__ == "__main__":
    main()
