"""torch-f64 restatement of the reference's JAX potentials.  TEST INFRASTRUCTURE ONLY.

Every energy function follows the cited reference function term by term; gradients come from torch
autograd (the reference: ``jax.grad`` of the same functions, tests/common.py:296-297).  The nonbonded
terms are evaluated over the explicit i<j pair list (the reference's own equivalent formulation is
``_nonbonded_clone`` in tests/test_jax_nonbonded.py:210-242), row-blocked so a 23k-atom box needs
O(block x N) memory rather than the dense N x N of ``nonbonded.nonbonded``.
"""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

SWITCH_CUTOFF = 1.2  # timemachine/potentials/nonbonded.py:36 (hard-coded, overrides user cutoff)


def _t(a, requires_grad=False):
    t = torch.as_tensor(np.asarray(a, dtype=np.float64))
    if requires_grad:
        t = t.clone().requires_grad_(True)
    return t


def delta_r(ri, rj, box_diag):
    """timemachine/potentials/jax_utils.py:37-44: diff -= L * floor(diff / L + 0.5)."""
    diff = ri - rj
    if box_diag is not None:
        diff = diff - box_diag * torch.floor(diff / box_diag + 0.5)
    return diff


def switch_fn(dij):
    """timemachine/potentials/nonbonded.py:23-39: cos(pi/2 (d/1.2)^8)^3 for d < 1.2 else 0."""
    f = torch.cos((torch.pi * (dij / SWITCH_CUTOFF) ** 8) / 2) ** 3
    return torch.where(dij < SWITCH_CUTOFF, f, torch.zeros_like(f))


def _pair_energies(dij, qij, sig_ij, eps_ij, beta, cutoff):
    """Per-pair (lj, es) following nonbonded.py:57-77 (lennard_jones, switched_direct_space_pme) and the
    masking of nonbonded.py:301-337 / 369-388: both terms are zero unless dij < cutoff; LJ is zero when
    eps_ij == 0."""
    within = dij < cutoff
    # guard the masked-out branch so autograd never sees inf * 0
    d_safe = torch.where(within, dij, torch.ones_like(dij))
    sig6 = (sig_ij / d_safe) ** 6
    lj = 4 * eps_ij * (sig6 * sig6 - sig6)
    lj = torch.where(within & (eps_ij != 0), lj, torch.zeros_like(lj))
    es = qij * torch.special.erfc(beta * d_safe) / d_safe * switch_fn(d_safe)
    es = torch.where(within, es, torch.zeros_like(es))
    return lj, es


def nonbonded_pairs_energy(conf, params, box, pairs, beta, cutoff, rescale=None):
    """Sum over explicit pairs; nonbonded.py:342-399 (nonbonded_on_specific_pairs).

    ``rescale`` [M,2] multiplies (es, lj) per pair, columns (charge, lj) as in the reference's
    ``rescale_mask``.  Returns a torch scalar."""
    if len(pairs) == 0:
        return conf.sum() * 0.0
    pairs = torch.as_tensor(np.asarray(pairs, dtype=np.int64))
    il, ir = pairs[:, 0], pairs[:, 1]
    box_diag = None if box is None else torch.diagonal(box)
    d3 = delta_r(conf[il], conf[ir], box_diag)
    dw = params[il, 3] - params[ir, 3]
    dij = torch.sqrt((d3 * d3).sum(-1) + dw * dw)
    qij = params[il, 0] * params[ir, 0]
    sig_ij = params[il, 1] + params[ir, 1]  # combining_rule_sigma, nonbonded.py:42-47
    eps_ij = params[il, 2] * params[ir, 2]  # combining_rule_epsilon, nonbonded.py:50-55
    lj, es = _pair_energies(dij, qij, sig_ij, eps_ij, beta, cutoff)
    if rescale is not None:
        rescale = _t(rescale)
        es = es * rescale[:, 0]
        lj = lj * rescale[:, 1]
    return lj.sum() + es.sum()


def nonbonded_all_pairs_energy(
    conf, params, box, beta, cutoff, atom_idxs=None, block=512, accumulate_backward=False, exclusion_idxs=None, scale_factors=None
):
    """All i<j pairs among ``atom_idxs`` (default: every atom); nonbonded.py:221-339.

    Exclusions are applied the way the reference applies them -- each excluded pair's (es, lj) is MULTIPLIED by
    (1 - scale) (convert_exclusions_to_rescale_masks, nonbonded.py:159-173, then nonbonded.py:337) -- not subtracted
    afterwards: with clashing excluded atoms a subtraction would cancel catastrophically in floating point.
    ``exclusion_idxs`` index into the full atom array (the caller filters them to ``atom_idxs``).

    With ``accumulate_backward`` the function calls ``.backward()`` per row block (memory O(block x N))
    and returns a python float; otherwise it returns a torch scalar that is still attached to the graph."""
    N = conf.shape[0]
    idx = torch.arange(N) if atom_idxs is None else torch.as_tensor(np.asarray(atom_idxs, dtype=np.int64))
    K = idx.shape[0]
    box_diag = None if box is None else torch.diagonal(box)
    # position of every atom inside `idx` (-1: not interacting)
    pos = np.full(N, -1, dtype=np.int64)
    pos[idx.numpy()] = np.arange(K)
    ex_lo = ex_hi = ex_q = ex_lj = None
    if exclusion_idxs is not None and len(exclusion_idxs):
        e = pos[np.asarray(exclusion_idxs, dtype=np.int64)]
        sf = np.asarray(scale_factors, dtype=np.float64).reshape(-1, 2)
        keep = (e >= 0).all(axis=1)
        e, sf = e[keep], sf[keep]
        ex_lo, ex_hi = np.minimum(e[:, 0], e[:, 1]), np.maximum(e[:, 0], e[:, 1])
        ex_q, ex_lj = 1.0 - sf[:, 0], 1.0 - sf[:, 1]
    total = 0.0 if accumulate_backward else conf.sum() * 0.0
    for r0 in range(0, K, block):
        r1 = min(r0 + block, K)
        ri = idx[r0:r1]
        cj = idx[r0:]  # columns >= row start; mask j > i below
        d3 = delta_r(conf[ri][:, None, :], conf[cj][None, :, :], box_diag)
        dw = params[ri, 3][:, None] - params[cj, 3][None, :]
        d2 = (d3 * d3).sum(-1) + dw * dw
        upper = (torch.arange(r0, r1)[:, None] < torch.arange(r0, K)[None, :])
        d2 = torch.where(upper, d2, torch.full_like(d2, 1e6))  # discard i>=j before sqrt (no 0-distance in graph)
        dij = torch.sqrt(d2)
        qij = params[ri, 0][:, None] * params[cj, 0][None, :]
        sig_ij = params[ri, 1][:, None] + params[cj, 1][None, :]
        eps_ij = params[ri, 2][:, None] * params[cj, 2][None, :]
        lj, es = _pair_energies(dij, qij, sig_ij, eps_ij, beta, cutoff)
        if ex_lo is not None:
            sel = (ex_lo >= r0) & (ex_lo < r1)
            if sel.any():
                mq = torch.ones_like(es)
                ml = torch.ones_like(lj)
                rr = torch.as_tensor(ex_lo[sel] - r0)
                cc = torch.as_tensor(ex_hi[sel] - r0)
                mq[rr, cc] = torch.as_tensor(ex_q[sel])  # later duplicates overwrite earlier ones, as in the reference
                ml[rr, cc] = torch.as_tensor(ex_lj[sel])
                es = es * mq
                lj = lj * ml
        u = lj.sum() + es.sum()
        if accumulate_backward:
            u.backward()
            total += float(u.detach())
        else:
            total = total + u
    return total


def filter_exclusions(atom_idxs, exclusion_idxs, scale_factors):
    """nonbonded.py:176-218 with update_idxs=False (what Nonbonded.to_gpu uses, potentials.py:135-136)."""
    keep = set(int(a) for a in atom_idxs)
    ei, sf = [], []
    for (i, j), s in zip(np.asarray(exclusion_idxs), np.asarray(scale_factors)):
        if int(i) in keep and int(j) in keep:
            ei.append((int(i), int(j)))
            sf.append(s)
    ei = np.array(ei, dtype=np.int32).reshape(-1, 2)
    sf = np.array(sf, dtype=np.float64).reshape(-1, 2)
    return ei, sf


def nonbonded_energy(conf, params, box, exclusion_idxs, scale_factors, beta, cutoff, atom_idxs=None, block=512):
    """nonbonded.py:221-339: all pairs with each excluded pair's (es, lj) multiplied by (1 - scale)."""
    return nonbonded_all_pairs_energy(
        conf, params, box, beta, cutoff, atom_idxs, block, exclusion_idxs=exclusion_idxs, scale_factors=scale_factors
    )


def harmonic_bond_energy(conf, params, bond_idxs):
    """timemachine/potentials/bonded.py:34-79."""
    if len(bond_idxs) == 0:
        return conf.sum() * 0.0
    b = torch.as_tensor(np.asarray(bond_idxs, dtype=np.int64))
    cij = conf[b[:, 0]] - conf[b[:, 1]]
    d2 = (cij * cij).sum(-1)
    kb, r0 = params[:, 0], params[:, 1]
    dij = torch.sqrt(torch.where(d2 == 0, torch.ones_like(d2), d2))
    dij = torch.where(d2 == 0, torch.zeros_like(dij), dij)
    e = torch.where(r0 == 0, kb / 2 * d2, kb / 2 * (dij - r0) ** 2)
    return e.sum()


def kahan_angle(ci, cj, ck, eps):
    """bonded.py:82-97."""
    rji = torch.cat([ci - cj, eps[:, None]], dim=-1)
    rjk = torch.cat([ck - cj, eps[:, None]], dim=-1)
    nji = torch.linalg.norm(rji, dim=-1, keepdim=True)
    njk = torch.linalg.norm(rjk, dim=-1, keepdim=True)
    y = torch.linalg.norm(njk * rji - nji * rjk, dim=-1)
    x = torch.linalg.norm(njk * rji + nji * rjk, dim=-1)
    return 2 * torch.atan2(y, x)


def harmonic_angle_energy(conf, params, angle_idxs):
    """bonded.py:100-138."""
    if len(angle_idxs) == 0:
        return conf.sum() * 0.0
    a = torch.as_tensor(np.asarray(angle_idxs, dtype=np.int64))
    ang = kahan_angle(conf[a[:, 0]], conf[a[:, 1]], conf[a[:, 2]], params[:, 2])
    return (params[:, 0] / 2 * (ang - params[:, 1]) ** 2).sum()


def signed_torsion_angle(ci, cj, ck, cl):
    """bonded.py:141-175."""
    rij = cj - ci
    rkj = cj - ck
    rkl = cl - ck
    n1 = torch.linalg.cross(rij, rkj)
    n2 = torch.linalg.cross(rkj, rkl)
    y = (torch.linalg.cross(n1, n2) * (rkj / torch.linalg.norm(rkj, dim=-1, keepdim=True))).sum(-1)
    x = (n1 * n2).sum(-1)
    return torch.atan2(y, x)


def periodic_torsion_energy(conf, params, torsion_idxs):
    """bonded.py:178-216."""
    if len(torsion_idxs) == 0:
        return conf.sum() * 0.0
    t = torch.as_tensor(np.asarray(torsion_idxs, dtype=np.int64))
    c = conf[:, :3]
    ang = signed_torsion_angle(c[t[:, 0]], c[t[:, 1]], c[t[:, 2]], c[t[:, 3]])
    return (params[:, 0] * (1 + torch.cos(params[:, 2] * ang - params[:, 1]))).sum()


# ---- interaction groups, precomputed pair lists, chiral restraints ----


def interaction_group_pairs(num_atoms, row_atom_idxs, col_atom_idxs=None):
    """All (row, col) pairs; nonbonded.py:449-481 (col defaults to the complement of the rows)."""
    rows = np.asarray(row_atom_idxs, dtype=np.int64)
    cols = np.setdiff1d(np.arange(num_atoms), rows) if col_atom_idxs is None else np.asarray(col_atom_idxs, dtype=np.int64)
    assert set(rows.tolist()).isdisjoint(cols.tolist())
    return np.stack([np.repeat(rows, len(cols)), np.tile(cols, len(rows))], 1)


def nonbonded_interaction_group_energy(conf, params, box, row_atom_idxs, col_atom_idxs, beta, cutoff):
    """nonbonded_interaction_groups (nonbonded.py:460-481): the pair-list energy over rows x cols."""
    pairs = interaction_group_pairs(conf.shape[0], row_atom_idxs, col_atom_idxs)
    return nonbonded_pairs_energy(conf, params, box, pairs, beta, cutoff)


def nonbonded_precomputed_energy(conf, params, box, pairs, beta, cutoff):
    """nonbonded_on_precomputed_pairs (nonbonded.py:403-446): params[pair] = (q_ij, sig_ij, eps_ij, w_offset_ij)."""
    if len(pairs) == 0:
        return conf.sum() * 0.0
    pairs = torch.as_tensor(np.asarray(pairs, dtype=np.int64))
    il, ir = pairs[:, 0], pairs[:, 1]
    box_diag = None if box is None else torch.diagonal(box)
    d3 = delta_r(conf[il], conf[ir], box_diag)
    dw = params[:, 3]
    dij = torch.sqrt((d3 * d3).sum(-1) + dw * dw)
    lj, es = _pair_energies(dij, params[:, 0], params[:, 1], params[:, 2], beta, cutoff)
    # nonbonded.py:443-444: q_ij == 0 selects the constant branch of a `where`, so d/dq_ij is 0 there (as in the kernel)
    es = torch.where(params[:, 0] != 0, es, torch.zeros_like(es))
    return lj.sum() + es.sum()


def _unit(v):
    return v / torch.linalg.norm(v, dim=-1, keepdim=True)


def chiral_atom_restraint_energy(conf, params, idxs):
    """chiral_restraints.py:9-37,60-72,107-117: k vol^2 where vol = (v0 x v1) . v2 > 0, else 0."""
    if len(idxs) == 0:
        return conf.sum() * 0.0
    idxs = torch.as_tensor(np.asarray(idxs, dtype=np.int64))
    xc, x1, x2, x3 = (conf[idxs[:, k]] for k in range(4))
    vol = (torch.cross(_unit(x1 - xc), _unit(x2 - xc), dim=-1) * _unit(x3 - xc)).sum(-1)
    return torch.where(vol > 0, params * vol * vol, torch.zeros_like(vol)).sum()


def chiral_bond_restraint_energy(conf, params, idxs, signs):
    """chiral_restraints.py:40-58,75-91,120-130: k vol^2 where sign * vol > 0, vol = (rij x rkj) . (rkj x rkl)."""
    if len(idxs) == 0:
        return conf.sum() * 0.0
    idxs = torch.as_tensor(np.asarray(idxs, dtype=np.int64))
    ci, cj, ck, cl = (conf[idxs[:, k]] for k in range(4))
    rij, rkj, rkl = _unit(cj - ci), _unit(cj - ck), _unit(cl - ck)
    vol = (torch.cross(rij, rkj, dim=-1) * torch.cross(rkj, rkl, dim=-1)).sum(-1)
    sgn = _t(np.asarray(signs, dtype=np.float64))
    return torch.where(vol * sgn > 0, params * vol * vol, torch.zeros_like(vol)).sum()


def flat_bottom_bond_energies(conf, params, box, bond_idxs):
    """_flat_bottom_bond_impl (bonded.py:219-232): per-bond (k/4) (r - r_max)^4 / (r - r_min)^4 outside [r_min, r_max]."""
    bond_idxs = torch.as_tensor(np.asarray(bond_idxs, dtype=np.int64))
    box_diag = None if box is None else torch.diagonal(box)
    d = delta_r(conf[bond_idxs[:, 0]], conf[bond_idxs[:, 1]], box_diag)
    r = torch.sqrt((d * d).sum(-1))
    k, r_min, r_max = params[:, 0], params[:, 1], params[:, 2]
    zero = torch.zeros_like(r)
    return (k / 4) * (torch.where(r > r_max, (r - r_max) ** 4, zero) + torch.where(r < r_min, (r - r_min) ** 4, zero))


def flat_bottom_bond_energy(conf, params, box, bond_idxs):
    return flat_bottom_bond_energies(conf, params, box, bond_idxs).sum()


def log_flat_bottom_bond_energy(conf, params, box, bond_idxs, beta):
    """bonded.py:245-253"""
    nrgs = flat_bottom_bond_energies(conf, params, box, bond_idxs)
    return (-torch.log(1 - torch.exp(-beta * nrgs))).sum() / beta


def centroid_restraint_energy(conf, group_a_idxs, group_b_idxs, kb, b0):
    """bonded.py:8-31"""
    a = torch.as_tensor(np.asarray(group_a_idxs, dtype=np.int64))
    b = torch.as_tensor(np.asarray(group_b_idxs, dtype=np.int64))
    dx = conf[a].mean(0) - conf[b].mean(0)
    d2 = (dx * dx).sum()
    if b0 == 0:
        return kb * d2
    return kb * (torch.sqrt(d2) - b0) ** 2


def value_and_grads(energy_fn, conf, params, *args, **kwargs) -> Tuple[float, np.ndarray, np.ndarray]:
    """u, du/dx, du/dp of ``energy_fn(conf, params, *args)`` by autograd (mirrors jax.grad(ref,(0,1)))."""
    x = _t(conf, True)
    p = _t(params, True)
    u = energy_fn(x, p, *args, **kwargs)
    gx, gp = torch.autograd.grad(u, (x, p), allow_unused=True)
    gx = np.zeros_like(conf) if gx is None else gx.numpy()
    gp = np.zeros_like(params) if gp is None else gp.numpy()
    return float(u.detach()), gx, gp


# ---- convenience front-ends with the reference dataclasses' call signature (conf, params, box) ----


def nonbonded(conf, params, box, exclusion_idxs, scale_factors, beta, cutoff, atom_idxs=None):
    return value_and_grads(
        lambda x, p: nonbonded_energy(x, p, _t(box), exclusion_idxs, scale_factors, beta, cutoff, atom_idxs),
        conf,
        params,
    )


def nonbonded_all_pairs(conf, params, box, beta, cutoff, atom_idxs=None):
    return value_and_grads(
        lambda x, p: nonbonded_all_pairs_energy(x, p, _t(box), beta, cutoff, atom_idxs), conf, params
    )


def nonbonded_pair_list(conf, params, box, pairs, rescale, beta, cutoff, negated=False):
    sign = -1.0 if negated else 1.0
    return value_and_grads(
        lambda x, p: sign * nonbonded_pairs_energy(x, p, _t(box), pairs, beta, cutoff, rescale), conf, params
    )


def harmonic_bond(conf, params, box, idxs):
    return value_and_grads(lambda x, p: harmonic_bond_energy(x, p, idxs), conf, params)


def harmonic_angle(conf, params, box, idxs):
    return value_and_grads(lambda x, p: harmonic_angle_energy(x, p, idxs), conf, params)


def periodic_torsion(conf, params, box, idxs):
    return value_and_grads(lambda x, p: periodic_torsion_energy(x, p, idxs), conf, params)


def nonbonded_forces_blocked(conf, params, box, beta, cutoff, exclusion_idxs=None, scale_factors=None, block=512):
    """du/dx (and u) of the full Nonbonded term for large N with O(block x N) memory: per-row-block
    backward passes.  Used by large-N parity tests."""
    x = _t(conf, True)
    p = _t(params, False)
    u = nonbonded_all_pairs_energy(
        x, p, _t(box), beta, cutoff, None, block, accumulate_backward=True, exclusion_idxs=exclusion_idxs, scale_factors=scale_factors
    )
    return u, x.grad.numpy()


def nonbonded_interaction_group(conf, params, box, row_atom_idxs, beta, cutoff, col_atom_idxs=None):
    return value_and_grads(
        lambda x, p: nonbonded_interaction_group_energy(x, p, _t(box), row_atom_idxs, col_atom_idxs, beta, cutoff), conf, params)


def nonbonded_pair_list_precomputed(conf, params, box, idxs, beta, cutoff):
    return value_and_grads(lambda x, p: nonbonded_precomputed_energy(x, p, _t(box), idxs, beta, cutoff), conf, params)


def chiral_atom_restraint(conf, params, box, idxs):
    return value_and_grads(lambda x, p: chiral_atom_restraint_energy(x, p, idxs), conf, params)


def chiral_bond_restraint(conf, params, box, idxs, signs):
    return value_and_grads(lambda x, p: chiral_bond_restraint_energy(x, p, idxs, signs), conf, params)


def flat_bottom_bond(conf, params, box, idxs):
    return value_and_grads(lambda x, p: flat_bottom_bond_energy(x, p, _t(box), idxs), conf, params)


def log_flat_bottom_bond(conf, params, box, idxs, beta):
    return value_and_grads(lambda x, p: log_flat_bottom_bond_energy(x, p, _t(box), idxs, beta), conf, params)


def centroid_restraint(conf, params, box, group_a_idxs, group_b_idxs, kb, b0):
    return value_and_grads(lambda x, p: centroid_restraint_energy(x, group_a_idxs, group_b_idxs, kb, b0), conf, np.zeros(0))
