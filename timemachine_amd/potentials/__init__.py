"""reference: timemachine/potentials/__init__.py (hot-path subset)."""
from .potential import (  # noqa: F401
    BoundGpuImplWrapper,
    BoundPotential,
    GpuImplWrapper,
    Potential,
)
from .potentials import (  # noqa: F401
    CentroidRestraint,
    ChiralAtomRestraint,
    ChiralBondRestraint,
    FanoutSummedPotential,
    FlatBottomBond,
    HarmonicAngle,
    HarmonicBond,
    LogFlatBottomBond,
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
