"""Golden vectors for the "next" rows of SURVEY.md section 8(f) and for the reference's edge cases, produced by running
the REFERENCE's own Python in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/generate_golden_next.py

  vv.npz        timemachine/integrator.py:153-199  VelocityVerletIntegrator.multiple_steps (fixed-point x, v) on flexible
                waters with bonded forces; oracle/integrator.py:velocity_verlet_device_model is asserted against it
  barostat.npz  timemachine/md/barostat/moves.py:39-83  CentroidRescaler.scale_centroids at the length scales the
                device proposal arithmetic (oracle/barostat.py:propose) produces from its Philox uniforms; the oracle is
                asserted equal to the reference modulo the home-box wrap the device adds (k_barostat.cuh)
  hrex.npz      timemachine/md/hrex.py:50-130  _run_neighbor_swaps fed fixed pair_idxs / uniforms; oracle/hrex.py and
                timemachine_amd.hrex.run_neighbor_swaps are asserted bitwise equal
  edge_*.npz    reference energies (nonbonded.nonbonded) for an orthorhombic box, the same system with every atom
                drifted by whole box vectors, and BASELINE config 1 (85 waters + 1 LJ atom = 256 atoms) in the 100 nm
                vacuum box and the 3.0 nm periodic box; gradients from oracle/ref_potentials.py after the same
                energy-equality + finite-difference checks as generate_golden.py
  edge_box_resize.npz / filter_exclusions.npz   the box + 1000 I case of tests/nonbonded/test_nonbonded.py:165-190; the
                reference's filter_exclusions outputs (the product's vectorised version is asserted equal)
  config4.npz   BASELINE config 4 shape: ~6.5k atoms, 4.0 nm box, 8 lambda windows with w_ligand = lambda * cutoff;
                reference energies u_k(x) for all 8 states on one frame (+ oracle gradients at one state)

The compiled extension is absent, so `timemachine.lib.custom_ops` resolves to the reference's own pure-Python stub file
(timemachine/lib/custom_ops.py); the one module constant the Python integrator reads from it, FIXED_EXPONENT
(= 0x1000000000, cpp/src/fixed_point.hpp:5 / wrap_kernels.cpp:2144), is set on that stub below.  Only data is written.
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

from timemachine.lib import custom_ops as _ref_custom_ops_stub  # noqa: E402

_ref_custom_ops_stub.FIXED_EXPONENT = 0x1000000000  # cpp/src/fixed_point.hpp:5 (the compiled module exports it)

from timemachine.integrator import VelocityVerletIntegrator as RefVelocityVerlet  # noqa: E402
from timemachine.md import hrex as ref_hrex  # noqa: E402
from timemachine.potentials import bonded as ref_bonded  # noqa: E402
from timemachine.potentials import nonbonded as ref_nonbonded  # noqa: E402

from oracle import barostat as obarostat  # noqa: E402
from oracle import hrex as ohrex  # noqa: E402
from oracle import integrator as ointegrator  # noqa: E402
from oracle import ref_potentials as rp  # noqa: E402
from oracle.fixed_point import float_to_fixed  # noqa: E402
from timemachine_amd import hrex as threx  # noqa: E402
from timemachine_amd import testsystems as ts  # noqa: E402

from generate_golden import check_fd, rel, strained  # noqa: E402  (same protocol as the hot-path goldens)

np.seterr(all="ignore")


def import_reference_centroid_rescaler():
    from timemachine.md.barostat.moves import CentroidRescaler

    return CentroidRescaler


# ---------------------------------------------------------------------------------------------------------------------
def gen_vv(rng, out):
    """Reference VelocityVerletIntegrator.multiple_steps on 12 flexible waters in vacuum, bonded forces only."""
    s = ts.build_water_box(12, 1.2, seed=11)
    box = np.eye(3) * 100.0
    x0 = s.coords.copy()
    x0[0::3] += rng.normal(size=(12, 3)) * 0.004  # strain the bonds and angles a little
    n = s.num_atoms
    v0 = rng.normal(size=(n, 3)) * 0.4
    dt, n_steps = 1.0e-3, 10

    # energy of the force function is the reference's (bonded.py); its gradient is the oracle's (checked against central
    # differences of the reference energy right here)
    def energy_ref(x):
        return float(ref_bonded.harmonic_bond(x, s.bond_params, box, s.bond_idxs)) + float(ref_bonded.harmonic_angle(x, s.angle_params, box, s.angle_idxs))

    def grad(x):
        _, gb, _ = rp.harmonic_bond(x, s.bond_params, box, s.bond_idxs)
        _, ga, _ = rp.harmonic_angle(x, s.angle_params, box, s.angle_idxs)
        return gb + ga

    check_fd("vv force", energy_ref, x0, grad(x0), rng, h=1e-6, tol=5e-6)
    intg = RefVelocityVerlet(lambda x: -grad(np.asarray(x)), s.masses, dt)
    # Context.multiple_steps(n) = initialize + n steps + finalize == the Python integrator's multiple_steps(n + 1)
    # (tests/test_velocity_verlet_integrator.py:128-136)
    ref_xs, ref_vs = intg.multiple_steps(x0, v0, n_steps=n_steps + 1)
    ref_xs, ref_vs = np.asarray(ref_xs), np.asarray(ref_vs)
    # the oracle's device model (double state, k_integrator.cuh:64-130) against the reference (fixed-point state)
    cbs = -dt / s.masses
    x_m, v_m = ointegrator.velocity_verlet_device_model(x0, v0, lambda x: float_to_fixed(grad(x)), cbs, dt, n_steps + 1)
    ex, ev = np.abs(x_m - ref_xs[-1]).max(), np.abs(v_m - ref_vs[-1]).max()
    print(f"  vv: oracle device model vs reference after {n_steps + 1} steps: |dx| {ex:.2e}  |dv| {ev:.2e}")
    assert ex < 1e-9 and ev < 1e-7, (ex, ev)  # fixed-point state quantisation (2^-36 per update) through 11 steps of stiff O-H dynamics
    np.savez_compressed(
        os.path.join(out, "vv.npz"), x0=x0, v0=v0, box=box, masses=s.masses, dt=dt, n_steps=n_steps, bond_idxs=s.bond_idxs,
        bond_params=s.bond_params, angle_idxs=s.angle_idxs, angle_params=s.angle_params, ref_xs=ref_xs, ref_vs=ref_vs,
    )


def gen_barostat(rng, out):
    CentroidRescaler = import_reference_centroid_rescaler()
    s = ts.add_chain_ligand(ts.build_water_box(40, 1.6, seed=5), 7, seed=3)
    N = s.num_atoms
    nw = (N - 7) // 3
    groups = [np.arange(3 * k, 3 * k + 3) for k in range(nw)] + [np.arange(N - 7, N)]
    x, box = s.coords.copy(), s.box.copy()
    x += rng.integers(-1, 2, size=(len(groups), 3)).repeat([len(g) for g in groups], axis=0) * np.diagonal(box)  # some molecules outside the home box
    center = np.diagonal(box) * 0.5
    seed, volume_scale = 77, 0.4
    resc = CentroidRescaler(groups)
    scales, x_refs = [], []
    for attempt in range(6):
        u1, _ = obarostat.attempt_uniforms(seed, attempt)
        # f64 arithmetic: the oracle's proposal IS the reference's centroid scaling, up to the wrap into the scaled home box
        x_p64, box_p64, (_, _, scale64) = obarostat.propose(x, box, groups, volume_scale, u1, real=np.float64)
        ref64 = np.asarray(resc.scale_centroids(x, center, scale64))
        shift = (x_p64 - ref64) / np.diagonal(box_p64)
        assert np.abs(shift - np.rint(shift)).max() < 1e-9, np.abs(shift - np.rint(shift)).max()
        for g in groups:
            assert np.all(np.rint(shift[g]) == np.rint(shift[g][0]))  # whole molecules are wrapped
        cent = np.array([x_p64[g].mean(0) for g in groups])
        assert np.all(cent >= -1e-9) and np.all(cent <= np.diagonal(box_p64) + 1e-9)
        # f32 arithmetic (what the device runs): reference scaling at the f32 length scale
        _, box_p32, (_, _, scale32) = obarostat.propose(x, box, groups, volume_scale, u1, real=np.float32)
        scales.append(scale32)
        x_refs.append(np.asarray(resc.scale_centroids(x, center, scale32)))
    print(f"  barostat: oracle.propose (f64) == reference CentroidRescaler modulo the home-box wrap; scales {np.round(scales, 5)}")
    np.savez_compressed(
        os.path.join(out, "barostat.npz"), x=x, box=box, group_sizes=np.array([len(g) for g in groups]), seed=seed, volume_scale=volume_scale,
        bond_idxs=s.bond_idxs, bond_params=s.bond_params, masses=s.masses, scales=np.array(scales), x_scaled=np.array(x_refs),
    )


def gen_hrex(rng, out):
    cases = {}
    for tag, n_states, n_attempts, hole in (("a", 8, 8**3, 0.0), ("b", 24, 4000, 0.6)):
        pairs = threx.neighbor_pairs(n_states)
        log_q = rng.normal(size=(n_states, n_states)) * 1.0
        if hole:  # unevaluated (replica, state) entries: U = +inf, log q = -inf (fe/free_energy.py:1187-1190)
            far = np.abs(np.arange(n_states)[:, None] - np.arange(n_states)[None, :]) > 4
            log_q = np.where(far, -np.inf, log_q)
        perm0 = rng.permutation(n_states)
        if hole:  # a state assignment the evaluated band can belong to: every replica within two states of its own index
            perm0 = np.arange(n_states)
            for k in rng.permutation(n_states - 1)[: n_states // 3]:
                perm0[[k, k + 1]] = perm0[[k + 1, k]]
        pair_idxs = rng.integers(0, len(pairs), n_attempts)
        uniforms = rng.random(n_attempts)
        import jax.numpy as jnp

        perm, proposed, accepted = ref_hrex._run_neighbor_swaps(jnp.array(perm0), jnp.array(pairs), jnp.array(log_q), pair_idxs, uniforms)
        perm, proposed, accepted = np.asarray(perm), np.asarray(proposed), np.asarray(accepted)
        p2, pr2, ac2 = threx.run_neighbor_swaps(perm0, pairs, log_q, pair_idxs, uniforms)
        assert np.array_equal(perm, p2) and np.array_equal(proposed, pr2) and np.array_equal(accepted, ac2)
        p3, pr3, ac3 = ohrex.run_moves(list(perm0), [tuple(p) for p in pairs], log_q, pair_idxs, uniforms)
        assert np.array_equal(perm, p3) and np.array_equal(proposed, pr3) and np.array_equal(accepted, ac3)
        assert int(accepted.sum()) > 0 and int(accepted.sum()) < int(proposed.sum())  # a chain that both accepts and rejects
        print(f"  hrex {tag}: {n_states} states, {n_attempts} attempts, {int(accepted.sum())} accepted; product + oracle == reference")
        cases.update({f"{tag}_perm0": perm0, f"{tag}_pairs": pairs, f"{tag}_log_q": log_q, f"{tag}_pair_idxs": pair_idxs, f"{tag}_uniforms": uniforms,
                      f"{tag}_perm": perm, f"{tag}_proposed": proposed, f"{tag}_accepted": accepted})
    np.savez_compressed(os.path.join(out, "hrex.npz"), **cases)


def _nb_case(name, x, p, box, excl, sc, beta, cutoff, rng, fd=True):
    u_ref = float(ref_nonbonded.nonbonded(x, p, box, excl, sc, beta, cutoff, runtime_validate=False))
    u, gx, gp = rp.nonbonded(x, p, box, excl, sc, beta, cutoff)
    assert rel(u, u_ref) < 1e-12, (name, u, u_ref)
    if fd:
        check_fd(name + " du_dx", lambda xx: float(ref_nonbonded.nonbonded(xx, p, box, excl, sc, beta, cutoff, runtime_validate=False)), x, gx, rng, n=6)
    print(f"  {name}: N={len(x)} u={u_ref:.6f}")
    return dict(x=x, params=p, box=box, exclusion_idxs=excl, scale_factors=sc, beta=beta, cutoff=cutoff, u=u_ref, du_dx=gx, du_dp=gp)


def gen_edges(rng, out):
    from generate_golden import random_nb_system

    beta, cutoff = 2.0, 1.2
    # orthorhombic box, every edge different (the reference supports ortholinear boxes only: wrap_kernels.cpp:51-78)
    n = 180
    box = np.diag([2.9, 3.7, 4.6])
    x, p, excl, sc = random_nb_system(rng, n, 1.0, cutoff, "half")
    x = (x * np.diagonal(box)).astype(np.float32).astype(np.float64)
    d = _nb_case("edge_ortho", x, p, box, excl, sc, beta, cutoff, rng)
    np.savez_compressed(os.path.join(out, "edge_ortho.npz"), **d)
    # the same atoms, each drifted by up to +-3 whole box vectors per axis: same energy, same forces
    shifts = rng.integers(-3, 4, size=(n, 3)) * np.diagonal(box)
    xd = x + shifts
    d2 = _nb_case("edge_drift", xd, p, box, excl, sc, beta, cutoff, rng, fd=False)
    assert rel(d2["u"], d["u"]) < 1e-9 and np.abs(d2["du_dx"] - d["du_dx"]).max() < 1e-6 * max(1.0, np.abs(d["du_dx"]).max())
    np.savez_compressed(os.path.join(out, "edge_drift.npz"), **d2)
    # BASELINE config 1: 85 waters + 1 neutral LJ atom = 256 atoms, vacuum (100 nm) and periodic (3.0 nm) boxes
    for tag, L in (("vacuum", 100.0), ("pbc", 3.0)):
        s = ts.config1_water_cluster(L)
        xs = strained(s, 1101)  # every bond stretched, every bonded atom pulled (generate_golden.strained)
        ps = s.nb_params.astype(np.float32).astype(np.float64)
        d = _nb_case(f"config1_{tag}", xs, ps, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, rng)
        for key, ref_fn, ora_fn, idx, prm in (
            ("bond", ref_bonded.harmonic_bond, rp.harmonic_bond, s.bond_idxs, s.bond_params),
            ("angle", ref_bonded.harmonic_angle, rp.harmonic_angle, s.angle_idxs, s.angle_params),
        ):
            u_ref = float(ref_fn(xs, prm, s.box, idx))
            u, gx, gp = ora_fn(xs, prm, s.box, idx)
            assert rel(u, u_ref) < 1e-12
            d.update({f"u_{key}": u_ref, f"du_dx_{key}": gx, f"du_dp_{key}": gp})
        np.savez_compressed(os.path.join(out, f"config1_{tag}.npz"), **d)


def gen_box_resize(rng, out):
    """tests/nonbonded/test_nonbonded.py:165-190: the same coordinates under box + 1000 I, then under the real box."""
    s = ts.build_water_box(300, 3.0, seed=4)
    x = s.coords.astype(np.float32).astype(np.float64)
    p = s.nb_params.astype(np.float32).astype(np.float64)
    d = dict(x=x, params=p, beta=s.beta, cutoff=s.cutoff)
    for tag, box in (("big", s.box + np.eye(3) * 1000.0), ("real", s.box)):
        u_ref = float(ref_nonbonded.nonbonded(x, p, box, np.zeros((0, 2), np.int32), np.zeros((0, 2)), s.beta, s.cutoff, runtime_validate=False))
        u, gx, gp = rp.nonbonded_all_pairs(x, p, box, s.beta, s.cutoff)
        assert rel(u, u_ref) < 1e-12, (tag, u, u_ref)
        d.update({f"box_{tag}": box, f"u_{tag}": u_ref, f"du_dx_{tag}": gx, f"du_dp_{tag}": gp})
        print(f"  box_resize {tag}: N={len(x)} u={u_ref:.6f}")
    np.savez_compressed(os.path.join(out, "edge_box_resize.npz"), **d)


def gen_filter(rng, out):
    """timemachine/potentials/nonbonded.py:176-218 filter_exclusions, both update_idxs settings, incl. an empty result."""
    from timemachine_amd.potentials import filter_exclusions

    n = 60
    excl = np.array([rng.choice(n, 2, replace=False) for _ in range(80)], dtype=np.int32)
    scales = rng.uniform(size=(80, 2))
    d = dict(exclusion_idxs=excl, scale_factors=scales)
    for k, atom_idxs in enumerate((rng.permutation(n)[:35].astype(np.int32), np.arange(n, dtype=np.int32), np.array([excl[0, 0]], dtype=np.int32))):
        d[f"atom_idxs_{k}"] = atom_idxs
        for upd in (False, True):
            ref_i, ref_s = ref_nonbonded.filter_exclusions(atom_idxs, excl, scales, update_idxs=upd)
            ref_i, ref_s = np.asarray(ref_i), np.asarray(ref_s)
            got_i, got_s = filter_exclusions(atom_idxs, excl, scales, update_idxs=upd)
            assert got_i.shape == ref_i.shape and np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s), (k, upd)
            assert got_i.dtype == np.int32
            d[f"idxs_{k}_{int(upd)}"] = ref_i.astype(np.int32)
            d[f"scales_{k}_{int(upd)}"] = ref_s.reshape(-1, 2)
    print("  filter_exclusions: product == reference on 3 atom sets x update_idxs in {False, True}")
    np.savez_compressed(os.path.join(out, "filter_exclusions.npz"), **d)


def gen_config4(rng, out):
    """~6.5k atoms in a 4.0 nm box (tests/test_benchmark.py:541 shape), 8 lambda windows, ligand w = lambda * cutoff."""
    lambdas = np.linspace(0.0, 1.0, 8)
    s0 = ts.config4_solvated_ligand(0.0)
    x = s0.coords.astype(np.float32).astype(np.float64)
    u_k = []
    params_k = []
    for lam in lambdas:
        s = ts.config4_solvated_ligand(float(lam))
        p = s.nb_params.astype(np.float32).astype(np.float64)
        params_k.append(p)
        u_ref = float(ref_nonbonded.nonbonded(x, p, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff, runtime_validate=False))
        u_or = float(rp.nonbonded_energy(rp._t(x), rp._t(p), rp._t(s.box), s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff))
        assert rel(u_or, u_ref) < 1e-12, (lam, u_or, u_ref)
        u_k.append(u_ref)
        print(f"  config4 lambda={lam:.3f}: N={s.num_atoms} u={u_ref:.4f}")
    k = 3
    _, gx, gp = rp.nonbonded(x, params_k[k], s0.box, s0.exclusion_idxs, s0.scale_factors, s0.beta, s0.cutoff)
    lig = np.arange(s0.num_water_atoms, s0.num_atoms)
    np.savez_compressed(
        os.path.join(out, "config4.npz"), lambdas=lambdas, x=x.astype(np.float32), u_k=np.array(u_k), grad_state=k,
        du_dx=gx, du_dp_ligand=gp[lig], n_ligand=len(lig),
    )


GENERATORS = {"vv": gen_vv, "barostat": gen_barostat, "hrex": gen_hrex, "edges": gen_edges, "box_resize": gen_box_resize, "filter": gen_filter,
              "config4": gen_config4}

if __name__ == "__main__":
    import time
    import zlib

    out = os.environ.get("TM_GOLDEN_OUT", HERE)
    which = sys.argv[1:] or list(GENERATORS)
    for name in which:
        # every generator draws from its OWN stream: any subset, in any order, reproduces the committed files
        rng = np.random.default_rng([20260928, zlib.crc32(name.encode())])
        t0 = time.time()
        GENERATORS[name](rng, out)
        print(f"{name}: {time.time() - t0:.1f} s")
    shutil.rmtree(_tmp, ignore_errors=True)
    print("done")
