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
    FanoutSummedPotential,
    HarmonicAngle,
    HarmonicBond,
    Nonbonded,
    NonbondedAllPairs,
    NonbondedExclusions,
    NonbondedPairList,
    PeriodicTorsion,
    SummedPotential,
    SummedPotentialGpuImplWrapper,
    filter_exclusions,
    make_summed_potential,
)
