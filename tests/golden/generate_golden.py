"""Generates tests/golden/*.npz by running the REFERENCE's own Python implementation of the hot path.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/generate_golden.py

What comes from where
  * energies (u)               : the reference's unmodified source -- timemachine/potentials/nonbonded.py
                                 (nonbonded, nonbonded_on_specific_pairs), bonded.py (harmonic_bond, harmonic_angle,
                                 periodic_torsion) and timemachine/integrator.py (LangevinIntegrator._step,
                                 langevin_coefficients) -- imported from /root/reference under a numpy-backed shim for
                                 the absent `jax` package (tests/golden/_jax_numpy_shim.py, materialised in a temp dir).
  * gradients (du_dx, du_dp)   : oracle/ref_potentials.py (torch-f64 autograd).  jax.grad is unavailable, so before
                                 anything is written this script asserts (i) oracle energy == reference energy to
                                 1e-12 relative and (ii) oracle gradients == central finite differences of the REFERENCE
                                 energy on sampled coordinates/parameters.
  * Hilbert keys / permutation : LUT from the reference's vendored C (oracle/_ref/libhilbert_ref.so, built by
                                 oracle/Makefile from cpp/src/vendored/hilbert.cpp where it lies), then the key/sort
                                 arithmetic of k_hilbert.cu restated in oracle/hilbert.py.
  * water.npy                  : a data file of the reference's own test-suite (tests/data/water.npy), copied verbatim.

Only data (inputs + expected outputs) is written; no reference source text.  TM_GOLDEN_OUT=<dir> writes there instead of
next to this script (tests/test_golden_regeneration.py regenerates every fixture into a temp dir and compares).
"""
import ctypes
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

from timemachine.integrator import LangevinIntegrator as RefLangevin  # noqa: E402
from timemachine.integrator import langevin_coefficients as ref_langevin_coefficients  # noqa: E402
from timemachine.potentials import bonded as ref_bonded  # noqa: E402
from timemachine.potentials import nonbonded as ref_nonbonded  # noqa: E402

from oracle import hilbert as ohilbert  # noqa: E402
from oracle import integrator as ointegrator  # noqa: E402
from oracle import ref_potentials as rp  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402

np.seterr(all="ignore")  # the reference's dense code divides by zero on the (masked) diagonal


def rel(a, b):
    return abs(a - b) / max(1.0, abs(b))


def check_fd(name, f_ref, x, g_oracle, rng, n=12, h=1e-5, tol=2e-6):
    """central finite differences of the REFERENCE energy vs the oracle gradient on n random entries"""
    flat = x.reshape(-1)
    worst = 0.0
    for k in rng.choice(flat.size, size=min(n, flat.size), replace=False):
        xp, xm = flat.copy(), flat.copy()
        xp[k] += h
        xm[k] -= h
        fd = (f_ref(xp.reshape(x.shape)) - f_ref(xm.reshape(x.shape))) / (2 * h)
        g = g_oracle.reshape(-1)[k]
        err = abs(fd - g) / max(1.0, abs(g))
        worst = max(worst, err)
        assert err < tol, (name, k, fd, g, err)
    print(f"  fd-check {name}: worst rel err {worst:.2e}")


def random_nb_system(rng, n, box_len, cutoff, w_mode):
    x = rng.uniform(0, box_len, (n, 3))
    # push apart pairs closer than 0.05 nm to keep energies representable
    params = np.stack(
        [
            (rng.uniform(size=n) - 0.5) * np.sqrt(138.935456),
            rng.uniform(size=n) / 5.0 / 2,
            np.sqrt(rng.uniform(size=n)),
            np.zeros(n),
        ],
        1,
    )
    params[rng.choice(n, n // 8, replace=False), 2] = 0.0  # some eps == 0 sites (LJ skipped)
    if w_mode == "random":
        params[:, 3] = rng.uniform(-cutoff, cutoff, n) * 0.5
    elif w_mode == "half":
        params[n // 2 :, 3] = 0.3 * cutoff
    E = n // 3
    excl = rng.choice(n, size=(E, 2), replace=False).astype(np.int32)
    scales = np.stack([rng.uniform(size=E), rng.uniform(size=E)], 1)
    scales[: E // 2] = 1.0  # fully excluded pairs as well as partial ones
    x = x.astype(np.float32).astype(np.float64)  # as compare_forces does (tests/common.py:288)
    params = params.astype(np.float32).astype(np.float64)
    return x, params, excl, scales


def gen_nonbonded(rng, out):
    beta, cutoff = 2.0, 1.2
    cases = {}
    for name, (n, L, w_mode) in {
        "nb_small_w0": (96, 3.0, "zero"),
        "nb_small_wrand": (160, 3.2, "random"),
        "nb_small_whalf": (131, 2.7, "half"),
    }.items():
        x, p, excl, sc = random_nb_system(rng, n, L, cutoff, w_mode)
        box = np.eye(3) * L
        u_ref = float(ref_nonbonded.nonbonded(x, p, box, excl, sc, beta, cutoff, runtime_validate=False))
        u, gx, gp = rp.nonbonded(x, p, box, excl, sc, beta, cutoff)
        assert rel(u, u_ref) < 1e-12, (name, u, u_ref)
        check_fd(name + " du_dx", lambda xx: float(ref_nonbonded.nonbonded(xx, p, box, excl, sc, beta, cutoff, runtime_validate=False)), x, gx, rng)
        check_fd(name + " du_dp", lambda pp: float(ref_nonbonded.nonbonded(x, pp, box, excl, sc, beta, cutoff, runtime_validate=False)), p, gp, rng, h=1e-6, tol=2e-5)
        # all pairs only, and the exclusion pair list on its own
        u_ap_ref = float(ref_nonbonded.nonbonded(x, p, box, np.zeros((0, 2), np.int32), np.zeros((0, 2)), beta, cutoff, runtime_validate=False))
        u_ap, gx_ap, gp_ap = rp.nonbonded_all_pairs(x, p, box, beta, cutoff)
        assert rel(u_ap, u_ap_ref) < 1e-12
        vdw, es = ref_nonbonded.nonbonded_on_specific_pairs(x, p, box, excl, beta, cutoff, sc)
        u_pl_ref = float(np.sum(vdw) + np.sum(es))
        u_pl, gx_pl, gp_pl = rp.nonbonded_pair_list(x, p, box, excl, sc, beta, cutoff)
        assert rel(u_pl, u_pl_ref) < 1e-12, (u_pl, u_pl_ref)
        # subset of atoms (atom_idxs)
        sub = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
        u_sub_ref = float(ref_nonbonded.nonbonded(x, p, box, excl, sc, beta, cutoff, runtime_validate=False, atom_idxs=sub))
        u_sub, gx_sub, gp_sub = rp.nonbonded(x, p, box, excl, sc, beta, cutoff, atom_idxs=sub)
        assert rel(u_sub, u_sub_ref) < 1e-12
        print(f"  {name}: u={u_ref:.6f} allpairs={u_ap_ref:.6f} pairlist={u_pl_ref:.6f} subset={u_sub_ref:.6f}")
        cases[name] = dict(
            x=x, params=p, box=box, exclusion_idxs=excl, scale_factors=sc, beta=beta, cutoff=cutoff,
            u=u_ref, du_dx=gx, du_dp=gp, u_all_pairs=u_ap_ref, du_dx_all_pairs=gx_ap, du_dp_all_pairs=gp_ap,
            u_pair_list=u_pl_ref, du_dx_pair_list=gx_pl, du_dp_pair_list=gp_pl,
            atom_idxs=sub, u_subset=u_sub_ref, du_dx_subset=gx_sub, du_dp_subset=gp_sub,
        )
    for name, d in cases.items():
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)


def gen_bonded(rng, out):
    n = 64
    x = rng.uniform(0, 1.5, (n, 3)).astype(np.float32).astype(np.float64)
    box = np.eye(3) * 100.0
    bonds = np.stack([rng.permutation(n)[:2] for _ in range(40)]).astype(np.int32)
    bp = np.stack([rng.uniform(100, 1000, 40), rng.uniform(0.1, 0.5, 40)], 1)
    bp[:5, 1] = 0.0  # b0 == 0 branch
    angles = np.stack([rng.permutation(n)[:3] for _ in range(50)]).astype(np.int32)
    ap = np.stack([rng.uniform(50, 500, 50), rng.uniform(0.5, 2.5, 50), np.zeros(50)], 1)
    ap[25:, 2] = rng.uniform(0.0, 0.2, 25)  # stabilised angles (eps > 0)
    tors = np.stack([rng.permutation(n)[:4] for _ in range(60)]).astype(np.int32)
    tp = np.stack([rng.uniform(1, 20, 60), rng.uniform(-np.pi, np.pi, 60), rng.integers(1, 7, 60).astype(float)], 1)
    d = dict(x=x, box=box, bond_idxs=bonds, bond_params=bp, angle_idxs=angles, angle_params=ap, torsion_idxs=tors, torsion_params=tp)
    for key, ref_fn, ora_fn, idx, prm in (
        ("bond", ref_bonded.harmonic_bond, rp.harmonic_bond, bonds, bp),
        ("angle", ref_bonded.harmonic_angle, rp.harmonic_angle, angles, ap),
        ("torsion", ref_bonded.periodic_torsion, rp.periodic_torsion, tors, tp),
    ):
        u_ref = float(ref_fn(x, prm, box, idx))
        u, gx, gp = ora_fn(x, prm, box, idx)
        assert rel(u, u_ref) < 1e-12, (key, u, u_ref)
        check_fd(key + " du_dx", lambda xx: float(ref_fn(xx, prm, box, idx)), x, gx, rng, h=1e-6, tol=5e-6)
        check_fd(key + " du_dp", lambda pp: float(ref_fn(x, pp, box, idx)), prm, gp, rng, h=1e-6, tol=5e-6)
        d.update({f"u_{key}": u_ref, f"du_dx_{key}": gx, f"du_dp_{key}": gp})
        print(f"  {key}: u={u_ref:.6f}")
    # the same system 50 nm away from the origin (exact in f64: the coordinates are f32 values below 2).  Bonded terms only see
    # differences, which the reference's kernels form in double before casting (k_harmonic_bond.cuh:27,
    # k_harmonic_angle.cuh:44-45, k_periodic_torsion.cuh:49-51): an f32 kernel that subtracted AFTER casting would lose
    # 4e-6 nm here.  Reference energies at the shifted coordinates; gradients are translation invariant.
    x_far = x + 50.0
    assert np.array_equal(x_far - 50.0, x)
    d["x_far"] = x_far
    for key, ref_fn, idx, prm in (
        ("bond", ref_bonded.harmonic_bond, bonds, bp), ("angle", ref_bonded.harmonic_angle, angles, ap), ("torsion", ref_bonded.periodic_torsion, tors, tp),
    ):
        u_far = float(ref_fn(x_far, prm, box, idx))
        assert rel(u_far, d[f"u_{key}"]) < 1e-12, (key, u_far, d[f"u_{key}"])
        d[f"u_{key}_far"] = u_far
    np.savez_compressed(os.path.join(out, "bonded.npz"), **d)


def strained(s, seed0, inflate=0.05, open_angles=0.05, sigma=0.0003, floor=100.0, angle_floor=0.01):
    """The system's coordinates with (i) every molecule inflated by 5 % about its centroid (every bond stretched), (ii) every
    angle opened by ~0.05 rad (the atoms moved along the sum of d(theta)/dx over the angle terms -- uniform inflation alone
    leaves angles where they are), (iii) N(0, 0.0003 nm) per component, rounded to f32 values.  Own generators: the shared
    stream is not touched.  Terms at their rest geometry carry forces ~ 0, against which a tolerance relative to the force norm
    (floor 1) says nothing: an f32 kernel's bond force has an absolute error of k * eps_f32 * r ~ 4.6e5 * 6e-8 * 0.1 = 3e-3
    kJ/mol/nm and its angle an error of ~4e-7 rad (3e-3 kJ/mol/nm as well for water) whatever the arithmetic.  The first seed
    for which EVERY bonded atom feels a bond force above `floor` kJ/mol/nm and every angle is off its rest value by more than
    `angle_floor` rad is taken, so 1e-4 of the norm (the f32 bar of the tests) lies above those errors everywhere."""
    import torch
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    n = s.num_atoms
    b = np.asarray(s.bond_idxs)
    _, mol = connected_components(coo_matrix((np.ones(len(b)), (b[:, 0], b[:, 1])), shape=(n, n)), directed=False)
    cen = np.zeros((mol.max() + 1, 3))
    np.add.at(cen, mol, s.coords)
    cen /= np.bincount(mol)[:, None]
    base = cen[mol] + (1.0 + inflate) * (s.coords - cen[mol])
    a = torch.as_tensor(np.asarray(s.angle_idxs, dtype=np.int64))

    def angles(xt):
        rji, rjk = xt[a[:, 0]] - xt[a[:, 1]], xt[a[:, 2]] - xt[a[:, 1]]
        return torch.acos((rji * rjk).sum(-1) / rji.norm(dim=-1) / rjk.norm(dim=-1))

    xt = torch.tensor(base, requires_grad=True)
    (grad,) = torch.autograd.grad(angles(xt).sum(), xt)
    grad = grad.numpy()
    # step length: the largest angle change a unit step produces is ~ |grad|^2 summed over an angle's three atoms
    per_angle = (grad[s.angle_idxs] ** 2).sum(axis=(1, 2))
    base = base + open_angles / np.median(per_angle) * grad
    theta0 = np.asarray(s.angle_params)[:, 1]
    bonded = np.unique(b)
    for seed in range(seed0, seed0 + 64):
        x = (base + np.random.default_rng(seed).normal(0.0, sigma, base.shape)).astype(np.float32).astype(np.float64)
        _, g, _ = rp.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs)
        off = np.abs(angles(torch.tensor(x)).numpy() - theta0)
        worst = (np.linalg.norm(g[bonded], axis=1).min(), off.min())
        if worst[0] >= floor and worst[1] >= angle_floor:
            return x
    raise RuntimeError(f"no seed strains every bond and angle enough (last: min bond force {worst[0]:.1f}, min angle offset {worst[1]:.4f})")


def gen_config2(rng, out):
    """~2 300-atom solvated ligand (BASELINE config 2), three lambda values."""
    for lamb in (0.0, 0.3, 1.0):
        s = ts.small_solvated_ligand(lamb=lamb)
        x = strained(s, 2202)
        p = s.nb_params.astype(np.float32).astype(np.float64)
        u_ref = float(ref_nonbonded.nonbonded(x, p, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, runtime_validate=False))
        u, gx, gp = rp.nonbonded(x, p, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)
        assert rel(u, u_ref) < 1e-12, (u, u_ref)
        ub_ref = float(ref_bonded.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs))
        ua_ref = float(ref_bonded.harmonic_angle(x, s.angle_params, s.box, s.angle_idxs))
        ut_ref = float(ref_bonded.periodic_torsion(x, s.torsion_params, s.box, s.torsion_idxs))
        ub, gxb, gpb = rp.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs)
        ua, gxa, gpa = rp.harmonic_angle(x, s.angle_params, s.box, s.angle_idxs)
        ut, gxt, gpt = rp.periodic_torsion(x, s.torsion_params, s.box, s.torsion_idxs)
        assert rel(ub, ub_ref) < 1e-12 and rel(ua, ua_ref) < 1e-12 and rel(ut, ut_ref) < 1e-12
        print(f"  config2 lambda={lamb}: N={s.num_atoms} u_nb={u_ref:.4f} bond={ub_ref:.4f} angle={ua_ref:.4f} torsion={ut_ref:.4f}")
        np.savez_compressed(
            os.path.join(out, f"config2_lambda{lamb:.1f}.npz"),
            lamb=lamb, x=x, nb_params=p, u_nonbonded=u_ref, du_dx_nonbonded=gx, du_dp_nonbonded=gp,
            u_bond=ub_ref, du_dx_bond=gxb, du_dp_bond=gpb, u_angle=ua_ref, du_dx_angle=gxa, du_dp_angle=gpa,
            u_torsion=ut_ref, du_dx_torsion=gxt, du_dp_torsion=gpt,
        )


def gen_integrator(rng, out):
    """12 BAOAB steps on a small water box, friction = 0 and friction = 1 with recorded noise (python reference)."""
    s = ts.build_water_box(85, 3.0, seed=3)
    x0 = s.coords.copy()
    n = s.num_atoms
    v0 = rng.normal(size=(n, 3)) * 0.3
    k = 50.0  # toy force field for the pure-integrator check: harmonic tether to x0 (force_fxn is arbitrary)

    def force(x):
        return -k * (x - x0) - 3.0 * (x - x0) ** 3

    d = dict(x0=x0, v0=v0, masses=s.masses, k=k)
    for friction in (0.0, 1.0):
        intg = RefLangevin(force, s.masses, 300.0, 2.5e-3, friction)
        ca, cb, cc = ref_langevin_coefficients(300.0, 2.5e-3, friction, s.masses)
        oca, ocb, occ = ointegrator.langevin_coefficients(300.0, 2.5e-3, friction, s.masses)
        assert np.array_equal(ca, oca) and np.array_equal(cb, ocb) and np.array_equal(cc, occ)
        x, v = x0.copy(), v0.copy()
        xo, vo = x0.copy(), v0.copy()
        noises, xs, vs = [], [], []
        for _ in range(12):
            noise = rng.normal(size=(n, 3))
            x, v = intg._step(x, v, noise)
            xo, vo = ointegrator.baoab_step(xo, vo, force(xo), noise, oca, ocb, occ, 2.5e-3)
            assert np.array_equal(np.asarray(x), xo) and np.array_equal(np.asarray(v), vo)
            noises.append(noise)
            xs.append(np.asarray(x).copy())
            vs.append(np.asarray(v).copy())
        d.update({f"noise_f{friction:.0f}": np.array(noises), f"xs_f{friction:.0f}": np.array(xs), f"vs_f{friction:.0f}": np.array(vs)})
        print(f"  integrator friction={friction}: oracle == reference bitwise over 12 steps")
    np.savez_compressed(os.path.join(out, "integrator.npz"), **d)


def gen_hilbert(rng, out):
    lib_path = os.path.join(REPO, "oracle", "_ref", "libhilbert_ref.so")
    lib = ctypes.CDLL(lib_path)
    lut_ref = np.zeros(128**3, dtype=np.uint32)
    lib.ref_hilbert_lut(128, 8, lut_ref.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(lut_ref, ohilbert.lut()), "oracle LUT != reference vendored hilbert_c2i"
    import hashlib

    sample_idx = rng.choice(128**3, 4096, replace=False)
    water = np.load(os.path.join(REF, "tests", "data", "water.npy"))[:, :3]
    box = np.eye(3) * (water.max(0) - water.min(0) + 0.1)
    keys = lut_ref[
        ((lambda b: b[:, 0].astype(np.int64) * 128 * 128 + b[:, 1].astype(np.int64) * 128 + b[:, 2])(
            ((water - np.diagonal(box) * np.floor(water / np.diagonal(box))) * (min(1 / np.diagonal(box)) * 127.0)).astype(np.uint32)))
    ]
    assert np.array_equal(keys, ohilbert.keys(water, box))
    perm = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(perm, ohilbert.sort_perm(water, box))
    np.savez_compressed(
        os.path.join(out, "hilbert.npz"), lut_sha256=hashlib.sha256(lut_ref.tobytes()).hexdigest(), lut_sample_idx=sample_idx,
        lut_sample_val=lut_ref[sample_idx], box=box, keys=keys, perm=perm,
    )
    shutil.copyfile(os.path.join(REF, "tests", "data", "water.npy"), os.path.join(out, "water.npy"))
    print(f"  hilbert: LUT sha256 {hashlib.sha256(lut_ref.tobytes()).hexdigest()[:16]}..., {len(np.unique(keys))} distinct keys / {len(keys)} atoms")


if __name__ == "__main__":
    rng = np.random.default_rng(20260927)
    out = os.environ.get("TM_GOLDEN_OUT", HERE)
    print("nonbonded"); gen_nonbonded(rng, out)
    print("bonded"); gen_bonded(rng, out)
    print("config 2"); gen_config2(rng, out)
    print("integrator"); gen_integrator(rng, out)
    print("hilbert"); gen_hilbert(rng, out)
    shutil.rmtree(_tmp, ignore_errors=True)
    print("done")
