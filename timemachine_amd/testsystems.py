"""Synthetic solvated systems for tests and bench.py (no OpenMM / RDKit in this image, so nothing is parsed from files).

The reference builds its systems with OpenMM (timemachine/md/builders.py, testsystems/dhfr.py) and converts them with
ff/handlers/openmm_deserializer.py:13-128; the parameter conventions below follow that deserializer:
  params[:, 0] = q * sqrt(ONE_4PI_EPS0)      params[:, 1] = sigma / 2      params[:, 2] = sqrt(epsilon)      params[:, 3] = w
  exclusion scale = fraction REMOVED (1.0 = fully excluded).
Water is TIP3P-like and flexible (the reference's DHFR benchmark uses flexible water too): standard TIP3P values.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .constants import ONE_4PI_EPS0

Q_O, Q_H = -0.834, 0.417
SIG_O, EPS_O = 0.315075, 0.635968  # nm, kJ/mol
K_OH, R_OH = 462750.4, 0.09572  # kJ/mol/nm^2, nm
K_HOH, THETA_HOH = 836.8, 1.82421813  # kJ/mol/rad^2, rad
M_O, M_H = 15.9994, 1.008
WATER_NUMBER_DENSITY = 33.4  # molecules / nm^3


@dataclass
class System:
    coords: np.ndarray  # [N,3] f64, nm
    box: np.ndarray  # [3,3]
    masses: np.ndarray  # [N]
    nb_params: np.ndarray  # [N,4]
    exclusion_idxs: np.ndarray  # [E,2] int32
    scale_factors: np.ndarray  # [E,2]
    bond_idxs: np.ndarray
    bond_params: np.ndarray
    angle_idxs: np.ndarray
    angle_params: np.ndarray
    torsion_idxs: np.ndarray
    torsion_params: np.ndarray
    beta: float = 2.0
    cutoff: float = 1.2
    num_water_atoms: int = 0

    @property
    def num_atoms(self):
        return self.coords.shape[0]


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a, b, c, d = q.T
    return np.stack(
        [
            np.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)], -1),
            np.stack([2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)], -1),
            np.stack([2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1),
        ],
        axis=1,
    )


def water_geometry():
    """O at the origin, H's in the xy-plane at the equilibrium geometry."""
    h = THETA_HOH / 2
    return np.array([[0.0, 0.0, 0.0], [R_OH * np.sin(h), R_OH * np.cos(h), 0.0], [-R_OH * np.sin(h), R_OH * np.cos(h), 0.0]])


def build_water_box(n_waters: int, box_length: float, seed: int = 2025, jitter: float = 0.02, hmr: bool = False,
                    cutoff: float = 1.2, beta: float = 2.0) -> System:
    """n_waters flexible TIP3P-like waters on a jittered cubic lattice with random orientations."""
    rng = np.random.default_rng(seed)
    m = int(np.ceil(n_waters ** (1 / 3)))
    spacing = box_length / m
    grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
    sel = rng.permutation(len(grid))[:n_waters]
    centers = (grid[sel] + 0.5) * spacing + rng.uniform(-jitter, jitter, (n_waters, 3))
    rots = _random_rotations(rng, n_waters)
    mol = water_geometry()
    coords = (centers[:, None, :] + np.einsum("nij,aj->nai", rots, mol)).reshape(-1, 3)
    N = 3 * n_waters
    o = np.arange(n_waters) * 3
    nb = np.zeros((N, 4))
    nb[o, 0] = Q_O * np.sqrt(ONE_4PI_EPS0)
    nb[o + 1, 0] = Q_H * np.sqrt(ONE_4PI_EPS0)
    nb[o + 2, 0] = Q_H * np.sqrt(ONE_4PI_EPS0)
    nb[o, 1] = SIG_O / 2
    nb[o + 1, 1] = 0.1 / 2  # sigma of an eps = 0 site is irrelevant; keep it finite and non-zero
    nb[o + 2, 1] = 0.1 / 2
    nb[o, 2] = np.sqrt(EPS_O)
    masses = np.tile([M_O, M_H, M_H], n_waters).astype(np.float64)
    if hmr:  # hydrogen-mass repartitioning as the reference benchmark does (tests/test_benchmark.py:207-214)
        masses = np.tile([M_O - 2 * M_H, 2 * M_H, 2 * M_H], n_waters).astype(np.float64)
    excl = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)], 1).reshape(-1, 2)
    bonds = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1)], 1).reshape(-1, 2)
    angles = np.stack([o + 1, o, o + 2], 1)
    return System(
        coords=coords,
        box=np.eye(3) * box_length,
        masses=masses,
        nb_params=nb,
        exclusion_idxs=excl.astype(np.int32),
        scale_factors=np.ones((len(excl), 2)),
        bond_idxs=bonds.astype(np.int32),
        bond_params=np.tile([K_OH, R_OH], (len(bonds), 1)).astype(np.float64),
        angle_idxs=angles.astype(np.int32),
        angle_params=np.tile([K_HOH, THETA_HOH, 0.0], (len(angles), 1)).astype(np.float64),
        torsion_idxs=np.zeros((0, 4), dtype=np.int32),
        torsion_params=np.zeros((0, 3)),
        beta=beta,
        cutoff=cutoff,
        num_water_atoms=N,
    )


def add_chain_ligand(sys: System, n_atoms: int = 20, lamb: float = 0.0, seed: int = 7) -> System:
    """Appends a synthetic n-atom chain "ligand" (bonds, angles, proper + 2 improper-like torsions; 1-2/1-3 exclusions
    fully removed, 1-4 scaled by 0.5; w = lamb * cutoff as in fe/topology.py:293) placed in the largest lattice void."""
    rng = np.random.default_rng(seed)
    L = sys.box[0, 0]
    # self-avoiding-ish random walk with 0.15 nm steps starting at the box centre
    pts = [np.array([L / 2, L / 2, L / 2])]
    direction = np.array([1.0, 0.0, 0.0])
    for _ in range(n_atoms - 1):
        d = direction + 0.8 * rng.normal(size=3)
        d /= np.linalg.norm(d)
        direction = d
        pts.append(pts[-1] + 0.15 * d)
    lig = np.array(pts)
    # drop waters that clash with the ligand (any atom within 0.25 nm)
    wat = sys.coords.reshape(-1, 3, 3)
    d = np.linalg.norm(wat[:, :, None, :] - lig[None, None, :, :], axis=-1)
    keep = np.nonzero(d.min(axis=(1, 2)) > 0.25)[0]
    nw = len(keep)
    base = build_water_box.__wrapped__ if hasattr(build_water_box, "__wrapped__") else None  # noqa: F841
    coords = np.concatenate([wat[keep].reshape(-1, 3), lig])
    No = 3 * nw
    idx = np.arange(n_atoms) + No

    def remap_water(arr, per):
        # water terms are laid out per molecule in build_water_box: keep the rows of the kept molecules, renumber
        arr = arr.reshape(-1, per, arr.shape[-1])[keep]
        old_first = keep * 3
        new_first = np.arange(nw) * 3
        return (arr - old_first[:, None, None] + new_first[:, None, None]).reshape(-1, arr.shape[-1])

    excl_w = remap_water(sys.exclusion_idxs, 3)
    bonds_w = remap_water(sys.bond_idxs, 2)
    angles_w = remap_water(sys.angle_idxs, 1)
    nb_w = sys.nb_params.reshape(-1, 3, 4)[keep].reshape(-1, 4)
    masses_w = sys.masses.reshape(-1, 3)[keep].reshape(-1)

    q = rng.normal(size=n_atoms) * 0.3
    q -= q.mean()
    nb_l = np.stack(
        [q * np.sqrt(ONE_4PI_EPS0), rng.uniform(0.1, 0.2, n_atoms), rng.uniform(0.2, 1.0, n_atoms), np.full(n_atoms, lamb * sys.cutoff)], 1
    )
    bonds_l = np.stack([idx[:-1], idx[1:]], 1)
    angles_l = np.stack([idx[:-2], idx[1:-1], idx[2:]], 1)
    tors_l = np.stack([idx[:-3], idx[1:-2], idx[2:-1], idx[3:]], 1)
    improper = np.array([[idx[1], idx[0], idx[2], idx[3]], [idx[5], idx[4], idx[6], idx[7]]]) if n_atoms >= 8 else np.zeros((0, 4), int)
    tors_all = np.concatenate([tors_l, improper])
    e12 = bonds_l
    e13 = np.stack([idx[:-2], idx[2:]], 1)
    e14 = np.stack([idx[:-3], idx[3:]], 1)
    excl_l = np.concatenate([e12, e13, e14])
    scales_l = np.concatenate([np.ones((len(e12) + len(e13), 2)), np.full((len(e14), 2), 0.5)])
    return System(
        coords=coords,
        box=sys.box.copy(),
        masses=np.concatenate([masses_w, np.full(n_atoms, 12.011)]),
        nb_params=np.concatenate([nb_w, nb_l]),
        exclusion_idxs=np.concatenate([excl_w, excl_l]).astype(np.int32),
        scale_factors=np.concatenate([np.ones((len(excl_w), 2)), scales_l]),
        bond_idxs=np.concatenate([bonds_w, bonds_l]).astype(np.int32),
        bond_params=np.concatenate([np.tile([K_OH, R_OH], (len(bonds_w), 1)), np.stack([rng.uniform(2e5, 3e5, len(bonds_l)), np.full(len(bonds_l), 0.15)], 1)]),
        angle_idxs=np.concatenate([angles_w, angles_l]).astype(np.int32),
        angle_params=np.concatenate([np.tile([K_HOH, THETA_HOH, 0.0], (len(angles_w), 1)), np.stack([rng.uniform(300, 600, len(angles_l)), rng.uniform(1.8, 2.1, len(angles_l)), np.zeros(len(angles_l))], 1)]),
        torsion_idxs=tors_all.astype(np.int32),
        torsion_params=np.stack([rng.uniform(1, 10, len(tors_all)), rng.uniform(0, np.pi, len(tors_all)), rng.integers(1, 4, len(tors_all)).astype(float)], 1),
        beta=sys.beta,
        cutoff=sys.cutoff,
        num_water_atoms=No,
    )


def dhfr_sized_water_box(seed: int = 2025, hmr: bool = True, cutoff: float = 1.2) -> System:
    """Config 3: 7 853 waters = 23 559 atoms in the 6.223 nm DHFR box (testsystems/data/5dfr_solv_equil.pdb:2)."""
    return build_water_box(7853, 6.223, seed=seed, hmr=hmr, cutoff=cutoff)


def small_solvated_ligand(lamb: float = 0.0, seed: int = 2025) -> System:
    """Config 2: ~760 waters + a 20-atom ligand in a 2.85 nm box (>= 2 * (1.2 + 0.1))."""
    return add_chain_ligand(build_water_box(772, 2.85, seed=seed), 20, lamb=lamb)


def config1_water_cluster(box_length: float, seed: int = 2025) -> System:
    """BASELINE config 1: 85 flexible waters (255 atoms) at liquid density + 1 neutral LJ atom = 256 atoms, as a
    1.37 nm cluster in the middle of a `box_length` box: 100.0 = the reference's "vacuum" box (tests/test_bonded.py:26),
    3.0 = the smallest periodic box its tests use (tests/test_md.py:44).  SURVEY.md section 8(d)."""
    side = (85 / WATER_NUMBER_DENSITY) ** (1 / 3)
    w = build_water_box(85, side, seed=seed)
    shift = 0.5 * (box_length - side)
    lj = np.array([[side + 0.25, 0.5 * side, 0.5 * side]])  # just outside one face of the cluster
    coords = np.concatenate([w.coords, lj]) + shift
    nb = np.concatenate([w.nb_params, [[0.0, 0.34 / 2, np.sqrt(0.5), 0.0]]])
    return System(
        coords=coords, box=np.eye(3) * box_length, masses=np.concatenate([w.masses, [39.948]]), nb_params=nb,
        exclusion_idxs=w.exclusion_idxs, scale_factors=w.scale_factors, bond_idxs=w.bond_idxs, bond_params=w.bond_params,
        angle_idxs=w.angle_idxs, angle_params=w.angle_params, torsion_idxs=w.torsion_idxs, torsion_params=w.torsion_params,
        beta=w.beta, cutoff=w.cutoff, num_water_atoms=w.num_atoms,
    )


def config4_solvated_ligand(lamb: float = 0.0, seed: int = 2025) -> System:
    """BASELINE config 4 shape: a 30-atom ligand in a 4.0 nm water box (~6.4k atoms; tests/test_benchmark.py:541),
    w_ligand = lamb * cutoff.  Windows differ only in the ligand's nonbonded parameters."""
    return add_chain_ligand(build_water_box(2138, 4.0, seed=seed), 30, lamb=lamb)


def config5_complex_sized(lamb: float = 0.0, seed: int = 2025) -> System:
    """BASELINE config 5 size: a 40-atom ligand in a 6.8 nm water box (~31k atoms, "30-60k-atom synthetic complex")."""
    return add_chain_ligand(build_water_box(10500, 6.8, seed=seed), 40, lamb=lamb)


def bound_potentials(sys: System, precision=np.float32, nblist_padding: float = 0.1):
    """[HarmonicBond, HarmonicAngle, (PeriodicTorsion), Nonbonded] bound to their parameters -- the shape of a
    reference state (fe/free_energy.py:614-657 packs the same list into one SummedPotential)."""
    from . import potentials as P

    bps = [
        P.HarmonicBond(sys.bond_idxs).bind(sys.bond_params),
        P.HarmonicAngle(sys.angle_idxs).bind(sys.angle_params),
    ]
    if len(sys.torsion_idxs):
        bps.append(P.PeriodicTorsion(sys.torsion_idxs).bind(sys.torsion_params))
    bps.append(
        P.Nonbonded(sys.num_atoms, sys.exclusion_idxs, sys.scale_factors, sys.beta, sys.cutoff, nblist_padding=nblist_padding).bind(sys.nb_params)
    )
    return bps
