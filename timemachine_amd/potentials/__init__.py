"""reference: timemachine/potentials/__init__.py (hot-path subset)."""
from .potential import (  # noqa: F401
    BoundGpuImplWrapper,
    BoundPotential,
    GpuImplWrapper,
    Potential,
    get_bound_potential_by_type,
    get_potential_by_type,
)
from .potentials import (  # noqa: F401
    ChiralAtomRestraint,
    ChiralBondRestraint,
    FanoutSummedPotential,
    HarmonicAngle,
    HarmonicBond,
    Nonbonded,
    NonbondedAllPairs,
    NonbondedExclusions,
    NonbondedInteractionGroup,
    NonbondedPairList,
    NonbondedPairListPrecomputed,
    PeriodicTorsion,
    SummedPotential,
    SummedPotentialGpuImplWrapper,
    filter_exclusions,
    make_summed_potential,
)
