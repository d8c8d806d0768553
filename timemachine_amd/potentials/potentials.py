"""The hot-path potentials as dataclasses with the reference's names, fields and field order
(reference: timemachine/potentials/potentials.py:17-31,94-161,203-304).
"""
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
from numpy.typing import NDArray

from ..lib import custom_ops
from .potential import BoundGpuImplWrapper, BoundPotential, GpuImplWrapper, Potential, Precision


@dataclass
class HarmonicBond(Potential):
    idxs: NDArray[np.int32]


@dataclass
class HarmonicAngle(Potential):
    idxs: NDArray[np.int32]


@dataclass
class PeriodicTorsion(Potential):
    idxs: NDArray[np.int32]


def filter_exclusions(atom_idxs, exclusion_idxs, scale_factors, update_idxs: bool = False):
    """The exclusions whose two atoms both belong to ``atom_idxs`` (with their scale factors); ``update_idxs`` renumbers
    the pairs to positions within ``atom_idxs``.  Same contract as timemachine/potentials/nonbonded.py:176-218."""
    atom_idxs = np.asarray(atom_idxs, dtype=np.int64).reshape(-1)
    pairs = np.asarray(exclusion_idxs, dtype=np.int64).reshape(-1, 2)
    scales = np.asarray(scale_factors)
    inside = np.isin(pairs, atom_idxs).all(axis=1) if len(atom_idxs) else np.zeros(len(pairs), dtype=bool)
    kept = pairs[inside]
    if update_idxs and len(kept):
        # position of each kept atom within atom_idxs (atom_idxs is unique, not necessarily sorted)
        order = np.argsort(atom_idxs, kind="stable")
        kept = order[np.searchsorted(atom_idxs[order], kept)]
    kept = kept.astype(np.int32)
    if not len(kept):
        kept = kept.reshape(-1)  # the reference returns a flat empty index array here (nonbonded.py:213); kept for drop-in equality
    return kept, scales[inside].reshape(-1, scales.shape[1])


@dataclass
class Nonbonded(Potential):
    """All pairs minus exclusions: on the GPU a FanoutSummedPotential([NonbondedAllPairs, NonbondedExclusions]) whose
    second member subtracts scale * pair in fixed point (reference: potentials.py:94-138)."""

    num_atoms: int
    exclusion_idxs: NDArray[np.int32]
    scale_factors: NDArray[np.float64]
    beta: float
    cutoff: float
    atom_idxs: Optional[NDArray[np.int32]] = None
    disable_hilbert_sort: bool = False
    nblist_padding: float = 0.1

    def _members(self):
        """(NonbondedAllPairs over the chosen atoms, NonbondedExclusions restricted to pairs inside that set)"""
        members_of = np.arange(self.num_atoms, dtype=np.int32) if self.atom_idxs is None else self.atom_idxs
        kept_idxs, kept_scales = filter_exclusions(members_of, self.exclusion_idxs, self.scale_factors)
        return (
            NonbondedAllPairs(self.num_atoms, self.beta, self.cutoff, self.atom_idxs, self.disable_hilbert_sort, self.nblist_padding),
            NonbondedExclusions(kept_idxs, kept_scales, self.beta, self.cutoff),
        )

    def to_gpu(self, precision: Precision) -> GpuImplWrapper:
        # both members take the SAME parameter array (fan-out); the second subtracts in fixed point what the first added
        return FanoutSummedPotential(list(self._members())).to_gpu(precision)


@dataclass
class NonbondedAllPairs(Potential):
    num_atoms: int
    beta: float
    cutoff: float
    atom_idxs: Optional[NDArray[np.int32]] = None
    disable_hilbert_sort: bool = False
    nblist_padding: float = 0.1

    def to_gpu(self, precision: Precision) -> GpuImplWrapper:
        # astuple() would deep-copy/convert the optional index array; pass fields explicitly in constructor order
        ctor = getattr(custom_ops, self._custom_ops_class_name(precision))
        return GpuImplWrapper(
            ctor(self.num_atoms, self.beta, self.cutoff, self.atom_idxs, self.disable_hilbert_sort, self.nblist_padding)
        )


@dataclass
class NonbondedInteractionGroup(Potential):
    """Row atoms x column atoms (reference: potentials.py:164-186).  ``col_atom_idxs=None``: all atoms not in the rows."""

    num_atoms: int
    row_atom_idxs: NDArray[np.int32]
    beta: float
    cutoff: float
    col_atom_idxs: Optional[NDArray[np.int32]] = None
    disable_hilbert_sort: bool = False
    nblist_padding: float = 0.1

    def to_gpu(self, precision: Precision) -> GpuImplWrapper:
        ctor = getattr(custom_ops, self._custom_ops_class_name(precision))
        return GpuImplWrapper(
            ctor(
                self.num_atoms, self.row_atom_idxs, self.beta, self.cutoff, self.col_atom_idxs, self.disable_hilbert_sort,
                self.nblist_padding,
            )
        )


@dataclass
class NonbondedPairListPrecomputed(Potential):
    """Pair list whose params are per PAIR: (q_ij, sig_ij, eps_ij, w_offset_ij), combining rules and scale factors already
    applied (reference: potentials.py:218-237)."""

    idxs: NDArray[np.int32]
    beta: float
    cutoff: float


@dataclass
class CentroidRestraint(Potential):
    """reference: potentials.py:49-57; no parameters (params of size 0)"""

    group_a_idxs: NDArray[np.int32]
    group_b_idxs: NDArray[np.int32]
    kb: float
    b0: float


@dataclass
class FlatBottomBond(Potential):
    """reference: potentials.py:77-82; params [B,3] = (k, r_min, r_max)"""

    idxs: NDArray[np.int32]


@dataclass
class LogFlatBottomBond(Potential):
    """reference: potentials.py:85-91"""

    idxs: NDArray[np.int32]
    beta: float


@dataclass
class ChiralAtomRestraint(Potential):
    """reference: potentials.py:60-65; params [R] force constants"""

    idxs: NDArray[np.int32]


@dataclass
class ChiralBondRestraint(Potential):
    """reference: potentials.py:68-74; params [R] force constants"""

    idxs: NDArray[np.int32]
    signs: NDArray[np.int32]


@dataclass
class NonbondedPairList(Potential):
    idxs: NDArray[np.int32]
    rescale_mask: NDArray[np.float64]
    beta: float
    cutoff: float


@dataclass
class NonbondedExclusions(Potential):
    idxs: NDArray[np.int32]
    rescale_mask: NDArray[np.float64]
    beta: float
    cutoff: float


@dataclass
class SummedPotential(Potential):
    potentials: Sequence[Potential]
    params_init: Sequence[NDArray]
    parallel: bool = True

    def __post_init__(self):
        if len(self.potentials) != len(self.params_init):
            raise ValueError("number of potentials != number of parameter arrays")

    def to_gpu(self, precision: Precision) -> "SummedPotentialGpuImplWrapper":
        impls = [p.to_gpu(precision).unbound_impl for p in self.potentials]
        sizes = [int(np.asarray(ps).size) for ps in self.params_init]
        return SummedPotentialGpuImplWrapper(custom_ops.SummedPotential(impls, sizes, self.parallel))

    def call_with_params_list(self, conf, params: Sequence[NDArray], box) -> float:
        params_flat = np.concatenate([np.asarray(ps).reshape(-1) for ps in params])
        return self(conf, params_flat, box)

    def bind_params_list(self, params: Sequence[NDArray]) -> BoundPotential["SummedPotential"]:
        params_flat = np.concatenate([np.asarray(ps).reshape(-1) for ps in params])
        return BoundPotential(self, params_flat)

    @property
    def params_shapes(self):
        return [np.asarray(ps).shape for ps in self.params_init]

    def unflatten_params(self, params):
        out, off = [], 0
        for shape in self.params_shapes:
            n = int(np.prod(shape))
            out.append(np.asarray(params)[off : off + n].reshape(shape))
            off += n
        return out


def make_summed_potential(bps: Sequence[BoundPotential]):
    potentials = [bp.potential for bp in bps]
    params = [bp.params for bp in bps]
    return SummedPotential(potentials, params).bind_params_list(params)


@dataclass
class SummedPotentialGpuImplWrapper(GpuImplWrapper):
    """Flattens parameter lists before calling the kernel wrapper (reference: potentials.py:275-291)."""

    def call_with_params_list(self, conf, params: Sequence[NDArray], box) -> float:
        params_flat = np.concatenate([np.asarray(ps).reshape(-1) for ps in params])
        return self(conf, params_flat, box)

    def bind_params_list(self, params: Sequence[NDArray]) -> BoundGpuImplWrapper:
        params_flat = np.concatenate([np.asarray(ps).reshape(-1) for ps in params])
        return BoundGpuImplWrapper(custom_ops.BoundPotential(self.unbound_impl, params_flat))


@dataclass
class FanoutSummedPotential(Potential):
    potentials: Sequence[Potential]
    parallel: bool = True

    def to_gpu(self, precision: Precision) -> GpuImplWrapper:
        impls = [p.to_gpu(precision).unbound_impl for p in self.potentials]
        return GpuImplWrapper(custom_ops.FanoutSummedPotential(impls, self.parallel))
