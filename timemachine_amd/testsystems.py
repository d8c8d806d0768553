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
    group_idxs: Optional[list] = None  # molecules (barostat groups); None: see molecule_groups()

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


# ---- a DHFR-shaped solute: a paraffin bundle with protein-like term counts --------------------------------------------
# amber99-like alkane parameters in the deserializer's conventions (HarmonicBond u = k/2 (r - r0)^2, so k = 2 K_amber)
R_CC, R_CH = 0.1526, 0.109
K_CC, K_CH = 259408.0, 284512.0  # 310 / 340 kcal/mol/A^2
THETA_TET = 1.9106332362490186  # acos(-1/3): every angle of the ideal geometry below
K_CCC, K_HCH, K_HCC = 334.72, 292.88, 418.4  # 40 / 35 / 50 kcal/mol/rad^2
SIG_CT, EPS_CT, SIG_HC, EPS_HC = 0.339967, 0.45773, 0.264953, 0.065689
K_TORSION_X = 0.650844  # X-CT-CT-X: 1.4 / 9 kcal/mol, n = 3, phase 0
SCALE_14 = (1.0 - 1.0 / 1.2, 0.5)  # fraction REMOVED of a 1-4 pair: charges scaled by 1/1.2, LJ by 1/2 (amber)


def _dihedral(p0, p1, p2, p3):
    """signed dihedral i-j-k-l with PeriodicTorsion's sign convention (timemachine/potentials/bonded.py:141-175): the two
    plane normals are (r_j - r_i) x (r_j - r_k) and (r_j - r_k) x (r_l - r_k), the sign is taken along r_j - r_k"""
    rij, rkj, rkl = p1 - p0, p1 - p2, p3 - p2
    n1, n2 = np.cross(rij, rkj), np.cross(rkj, rkl)
    return np.arctan2(np.dot(np.cross(n1, n2), rkj) / np.linalg.norm(rkj), np.dot(n1, n2))


def paraffin_chain(n_carbons: int):
    """One all-trans C_n H_(2n+2) chain along x at its force-field minimum (every angle acos(-1/3)): coordinates,
    carbon / hydrogen index lists and the bonded topology.  Atom order: C0, its H's, C1, its H's, ..."""
    half = THETA_TET / 2
    step, rise = R_CC * np.sin(half), R_CC * np.cos(half)
    coords, carbons, h_of = [], [], []
    for k in range(n_carbons):
        s = 1.0 if k % 2 == 0 else -1.0
        c = np.array([k * step, 0.5 * s * rise, 0.0])
        carbons.append(len(coords))
        coords.append(c)
        hs = []
        for sign in (1.0, -1.0):  # the two H's every carbon has: away from the chain axis, above and below the zigzag plane
            hs.append(len(coords))
            coords.append(c + R_CH * np.array([0.0, s * np.cos(half), sign * np.sin(half)]))
        if k == 0 or k == n_carbons - 1:  # methyl ends: the third H sits where the next carbon would be
            hs.append(len(coords))
            coords.append(c + R_CH * np.array([(-1.0 if k == 0 else 1.0) * np.sin(half), -s * np.cos(half), 0.0]))
        h_of.append(hs)
    coords = np.array(coords)
    n = len(coords)
    nbrs = [[] for _ in range(n)]
    bonds = []
    for k in range(n_carbons):
        for h in h_of[k]:
            bonds.append((carbons[k], h))
        if k + 1 < n_carbons:
            bonds.append((carbons[k], carbons[k + 1]))
    for i, j in bonds:
        nbrs[i].append(j)
        nbrs[j].append(i)
    angles = [(a, c, b) for c in range(n) for ia, a in enumerate(nbrs[c]) for b in nbrs[c][ia + 1 :]]
    torsions = [(a, i, j, b) for (i, j) in bonds if len(nbrs[i]) > 1 and len(nbrs[j]) > 1 for a in nbrs[i] if a != j for b in nbrs[j] if b != i]
    return coords, carbons, h_of, bonds, angles, torsions


def dhfr_shaped_box(seed: int = 2025, hmr: bool = True, cutoff: float = 1.2, beta: float = 2.0) -> System:
    """Config 3 with DHFR's SHAPE (testsystems/dhfr.py:9-23: 2 489 protein atoms + 21 069 water atoms in a 6.223 nm box, every
    bonded term kind, 1-4 exclusions with partial scales): 7 023 flexible TIP3P-like waters + a 2 490-atom solute = 23 559
    atoms.  The solute is a bundle of 30 all-trans C27H56 chains (a paraffin crystallite, 6 x 5 chains) with amber99-like
    alkane parameters: 2 460 bonds, 4 860 angles, 8 610 PeriodicTorsion terms (7 020 X-C-C-X n=3 terms, 1 440 C-C-C-C n=2 / n=1
    terms, 150 improper-form terms around carbons), 2 460 + 4 860 fully excluded 1-2 / 1-3 pairs and 7 020 1-4 pairs scaled
    by (1/1.2, 1/2) -- DHFR itself has 2 523 / 4 547 / ~6 700 + 418.  Waters first, solute last (num_water_atoms)."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    L = 6.223
    n_waters, ny, nz, nc = 7023, 6, 5, 27
    dy, dz = 0.48, 0.44
    c_xyz, carbons, h_of, bonds, angles, torsions = paraffin_chain(nc)
    per_chain = len(c_xyz)
    assert ny * nz * per_chain == 2490
    length = c_xyz[:, 0].max() - c_xyz[:, 0].min()
    origin = np.array([0.5 * (L - length), 0.5 * (L - (ny - 1) * dy), 0.5 * (L - (nz - 1) * dz)])
    sol_xyz, sol_bonds, sol_angles, sol_tors, sol_tparams, sol_is_c, sol_q = [], [], [], [], [], [], []
    c_set = set(carbons)
    for iy in range(ny):
        for iz in range(nz):
            off = len(sol_xyz) * per_chain
            shift = origin + np.array([0.0, iy * dy, iz * dz])
            sol_xyz.append(c_xyz + shift)
            sol_bonds += [(i + off, j + off) for i, j in bonds]
            sol_angles += [(a + off, c + off, b + off) for a, c, b in angles]
            for a, i, j, b in torsions:
                sol_tors.append((a + off, i + off, j + off, b + off))
                sol_tparams.append((K_TORSION_X, 0.0, 3.0))
                if a in c_set and b in c_set:  # C-C-C-C: amber's extra n = 2 and n = 1 terms (0.25 / 0.2 kcal/mol, phase pi)
                    sol_tors += [(a + off, i + off, j + off, b + off)] * 2
                    sol_tparams += [(1.046, np.pi, 2.0), (0.8368, np.pi, 1.0)]
            for k in range(5, nc - 1, 5):  # improper-form terms (central atom third), minimum at the built geometry
                quad = (carbons[k - 1], carbons[k + 1], carbons[k], h_of[k][0])
                phi0 = _dihedral(*[c_xyz[q] for q in quad])
                sol_tors.append(tuple(q + off for q in quad))
                sol_tparams.append((4.6024, 2.0 * phi0 - np.pi, 2.0))  # 1.1 kcal/mol, n = 2: k (1 + cos(2 phi - phase)) minimal at phi0
            is_c = np.zeros(per_chain, dtype=bool)
            is_c[carbons] = True
            q = np.where(is_c, -0.12, 0.06)
            q[[carbons[0], carbons[-1]]] = -0.18  # methyl carbons: neutral CH3 / CH2 groups (OPLS-like)
            sol_is_c.append(is_c)
            sol_q.append(q)
    sol_xyz = np.concatenate(sol_xyz)
    sol_is_c, sol_q = np.concatenate(sol_is_c), np.concatenate(sol_q)
    n_sol = len(sol_xyz)

    # water: a jittered 21^3 lattice, sites closer than 0.30 nm to any solute atom dropped, 7 023 of the rest kept
    m = 21
    spacing = L / m
    grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centers = (grid + 0.5) * spacing
    free = np.array([len(hit) == 0 for hit in cKDTree(sol_xyz).query_ball_point(centers, 0.30)])
    centers = centers[free]
    assert len(centers) >= n_waters, (len(centers), n_waters)
    centers = centers[rng.permutation(len(centers))[:n_waters]] + rng.uniform(-0.02, 0.02, (n_waters, 3))
    rots = _random_rotations(rng, n_waters)
    wat_xyz = (centers[:, None, :] + np.einsum("nij,aj->nai", rots, water_geometry())).reshape(-1, 3)
    Nw = 3 * n_waters
    o = np.arange(n_waters) * 3
    nb_w = np.zeros((Nw, 4))
    nb_w[o, 0], nb_w[o + 1, 0], nb_w[o + 2, 0] = Q_O * np.sqrt(ONE_4PI_EPS0), Q_H * np.sqrt(ONE_4PI_EPS0), Q_H * np.sqrt(ONE_4PI_EPS0)
    nb_w[o, 1], nb_w[o + 1, 1], nb_w[o + 2, 1] = SIG_O / 2, 0.05, 0.05
    nb_w[o, 2] = np.sqrt(EPS_O)
    excl_w = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1), np.stack([o + 1, o + 2], 1)], 1).reshape(-1, 2)
    bonds_w = np.stack([np.stack([o, o + 1], 1), np.stack([o, o + 2], 1)], 1).reshape(-1, 2)
    angles_w = np.stack([o + 1, o, o + 2], 1)

    sol_bonds, sol_angles, sol_tors = np.array(sol_bonds) + Nw, np.array(sol_angles) + Nw, np.array(sol_tors) + Nw
    sol_tparams = np.array(sol_tparams)
    nb_s = np.stack([sol_q * np.sqrt(ONE_4PI_EPS0), np.where(sol_is_c, SIG_CT, SIG_HC) / 2, np.sqrt(np.where(sol_is_c, EPS_CT, EPS_HC)), np.zeros(n_sol)], 1)
    cc = sol_is_c[sol_bonds[:, 0] - Nw] & sol_is_c[sol_bonds[:, 1] - Nw]
    bond_p = np.stack([np.where(cc, K_CC, K_CH), np.where(cc, R_CC, R_CH)], 1)
    n_c_ends = sol_is_c[sol_angles[:, 0] - Nw].astype(int) + sol_is_c[sol_angles[:, 2] - Nw].astype(int)
    angle_p = np.stack([np.choose(n_c_ends, [K_HCH, K_HCC, K_CCC]), np.full(len(sol_angles), THETA_TET), np.zeros(len(sol_angles))], 1)
    e13 = sol_angles[:, [0, 2]]
    proper = sol_tparams[:, 2] == 3.0  # one 1-4 pair per X-C-C-X quadruple (the extra C-C-C-C and improper terms add none)
    e14 = sol_tors[proper][:, [0, 3]]
    excl_s = np.concatenate([sol_bonds, e13, e14])
    scales_s = np.concatenate([np.ones((len(sol_bonds) + len(e13), 2)), np.tile(SCALE_14, (len(e14), 1))])
    assert len(np.unique(np.sort(excl_s, axis=1), axis=0)) == len(excl_s)

    m_w = [M_O - 2 * M_H, 2 * M_H, 2 * M_H] if hmr else [M_O, M_H, M_H]
    masses_s = np.where(sol_is_c, 12.011, 1.008)
    if hmr:  # every hydrogen twice as heavy, its carbon lighter by as much (as for the waters; tests/test_benchmark.py:207-214)
        n_h = np.zeros(n_sol)
        np.add.at(n_h, sol_bonds[~cc][:, 0] - Nw, 1.0)  # (C, H) bonds list the carbon first
        masses_s = np.where(sol_is_c, 12.011 - 1.008 * n_h, 2 * 1.008)
    return System(
        coords=np.concatenate([wat_xyz, sol_xyz]),
        box=np.eye(3) * L,
        masses=np.concatenate([np.tile(m_w, n_waters), masses_s]).astype(np.float64),
        nb_params=np.concatenate([nb_w, nb_s]),
        exclusion_idxs=np.concatenate([excl_w, excl_s]).astype(np.int32),
        scale_factors=np.concatenate([np.ones((len(excl_w), 2)), scales_s]),
        bond_idxs=np.concatenate([bonds_w, sol_bonds]).astype(np.int32),
        bond_params=np.concatenate([np.tile([K_OH, R_OH], (len(bonds_w), 1)), bond_p]),
        angle_idxs=np.concatenate([angles_w, sol_angles]).astype(np.int32),
        angle_params=np.concatenate([np.tile([K_HOH, THETA_HOH, 0.0], (len(angles_w), 1)), angle_p]),
        torsion_idxs=sol_tors.astype(np.int32),
        torsion_params=sol_tparams,
        beta=beta,
        cutoff=cutoff,
        num_water_atoms=Nw,
        group_idxs=[list(range(3 * i, 3 * i + 3)) for i in range(n_waters)]
        + [list(range(Nw + c * per_chain, Nw + (c + 1) * per_chain)) for c in range(ny * nz)],
    )


def molecule_groups(sys: System):
    """atom index lists of the molecules (what a MonteCarloBarostat rescales as rigid units): waters, then whatever the
    builder recorded for the solute (one group for everything else otherwise)"""
    if sys.group_idxs is not None:
        return sys.group_idxs
    nw = sys.num_water_atoms // 3
    groups = [list(range(3 * i, 3 * i + 3)) for i in range(nw)]
    if sys.num_atoms > sys.num_water_atoms:
        groups.append(list(range(sys.num_water_atoms, sys.num_atoms)))
    return groups


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


# ---- the composition the reference's RBFE / AHFE states have (timemachine/fe/system.py:133-146) -------------------------------
def rbfe_shaped_state(sys: System, n_ligand: int, nblist_padding: float = 0.1, env_charge_scale: Optional[float] = None):
    """``HostGuestSystem.get_U_fns()`` for a System whose LAST ``n_ligand`` atoms are the ligand (add_chain_ligand): the list the
    reference's single-topology code builds (fe/single_topology.py:2100-2154, host and guest combined) and packs into every
    RBFE window -- in the dataclass's field order, ``chiral_bond`` left out as the reference leaves it out (fe/system.py:100-107):

        bond, angle, proper, improper                       host + ligand terms concatenated
        chiral_atom                                         ligand only
        nonbonded_pair_list   NonbondedPairListPrecomputed  ligand-ligand pairs, combining rules and 1-4 scales folded in
        nonbonded_all_pairs   Nonbonded(atom_idxs=host)     host-host; parameters [host; zeros for the ligand] (:1984-2008)
        nonbonded_ixn_group   NonbondedInteractionGroup     ligand rows x host columns; parameters [host; ligand], w_ligand from
                                                            lambda (:2010-2055); ``env_charge_scale`` rescales the host charges
                                                            here only, as an environment BCC handle does (:2040-2043)

    Returns [(potential dataclass, parameter array)]: bind them one by one for a Context (fe/free_energy.py:614-657 packs the same
    list into one SummedPotential), or hand the two lists to potentials.SummedPotential."""
    from . import potentials as P

    N = sys.num_atoms
    n_host = N - n_ligand
    lig = np.arange(n_host, N, dtype=np.int32)
    host = np.arange(n_host, dtype=np.int32)
    # proper / improper: add_chain_ligand appends two improper-form terms after the ligand's proper torsions
    n_improper = 2 if (n_ligand >= 8 and len(sys.torsion_idxs) >= 2) else 0
    n_tors = len(sys.torsion_idxs)
    proper_idxs, proper_params = sys.torsion_idxs[: n_tors - n_improper], sys.torsion_params[: n_tors - n_improper]
    improper_idxs, improper_params = sys.torsion_idxs[n_tors - n_improper :], sys.torsion_params[n_tors - n_improper :]
    # chiral centres along the chain: (centre, three others), restrained at the reference's strength (fe/chiral_utils.py)
    centres = lig[1:-2:4]
    chiral_idxs = np.stack([centres, centres - 1, centres + 1, centres + 2], 1).astype(np.int32)
    chiral_params = np.full(len(chiral_idxs), 1000.0)
    # ligand-ligand pairs with the exclusion scales folded in (value = fraction REMOVED); fully excluded pairs dropped
    in_lig = (sys.exclusion_idxs >= n_host).all(axis=1)
    removed = {(int(i), int(j)): sc for (i, j), sc in zip(np.sort(sys.exclusion_idxs[in_lig], axis=1), sys.scale_factors[in_lig])}
    pairs, pair_params = [], []
    for a in range(n_host, N):
        for b in range(a + 1, N):
            sq, slj = removed.get((a, b), (0.0, 0.0))
            if sq == 1.0 and slj == 1.0:
                continue
            pa, pb = sys.nb_params[a], sys.nb_params[b]
            pairs.append((a, b))
            pair_params.append(((1.0 - sq) * pa[0] * pb[0], pa[1] + pb[1], (1.0 - slj) * pa[2] * pb[2], 0.0))
    pairs = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    pair_params = np.array(pair_params, dtype=np.float64).reshape(-1, 4)
    # host-host: exclusions among host atoms, ligand rows of the parameters zeroed (the choice is arbitrary: never read)
    in_host = (sys.exclusion_idxs < n_host).all(axis=1)
    host_params = sys.nb_params.copy()
    host_params[n_host:] = 0.0
    ixn_params = sys.nb_params.copy()
    if env_charge_scale is not None:
        ixn_params[:n_host, 0] *= env_charge_scale
    return [
        (P.HarmonicBond(sys.bond_idxs), sys.bond_params),
        (P.HarmonicAngle(sys.angle_idxs), sys.angle_params),
        (P.PeriodicTorsion(proper_idxs), proper_params),
        (P.PeriodicTorsion(improper_idxs), improper_params),
        (P.ChiralAtomRestraint(chiral_idxs), chiral_params),
        (P.NonbondedPairListPrecomputed(pairs, sys.beta, sys.cutoff), pair_params),
        (P.Nonbonded(N, sys.exclusion_idxs[in_host], sys.scale_factors[in_host], sys.beta, sys.cutoff, atom_idxs=host, nblist_padding=nblist_padding), host_params),
        (P.NonbondedInteractionGroup(N, lig, sys.beta, sys.cutoff, col_atom_idxs=host, nblist_padding=nblist_padding), ixn_params),
    ]


def rbfe_bound_potentials(sys: System, n_ligand: int, nblist_padding: float = 0.1, env_charge_scale: Optional[float] = None):
    """rbfe_shaped_state, each potential bound to its parameters (the `bps` of a Context)"""
    return [pot.bind(np.asarray(prm, dtype=np.float64)) for pot, prm in rbfe_shaped_state(sys, n_ligand, nblist_padding, env_charge_scale)]
