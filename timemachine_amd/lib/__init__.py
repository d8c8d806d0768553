"""Thin dataclasses whose ``.impl()`` constructs ``custom_ops`` objects (reference: timemachine/lib/__init__.py:12-62)."""
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np
from numpy.typing import NDArray

from . import custom_ops


@dataclass
class LangevinIntegrator:
    """reference: timemachine/lib/__init__.py:12-21 (same field order, same impl() call)."""

    temperature: float
    dt: float
    friction: float
    masses: NDArray[np.float64]
    seed: int

    def impl(self):
        return custom_ops.LangevinIntegrator(self.masses, self.temperature, self.dt, self.friction, self.seed)


@dataclass
class VelocityVerletIntegrator:
    """reference: timemachine/lib/__init__.py:24-37 (cbs = -dt / masses, computed in __post_init__)."""

    dt: float
    masses: NDArray[np.float64]

    cbs: NDArray[np.float64] = field(init=False)

    def __post_init__(self):
        cb = self.dt / np.asarray(self.masses, dtype=np.float64)
        cb *= -1
        self.cbs = cb

    def impl(self):
        return custom_ops.VelocityVerletIntegrator(self.dt, self.cbs)


@dataclass
class MonteCarloBarostat:
    """reference: timemachine/lib/__init__.py:40-62 (same fields; impl(bound_potentials) takes custom_ops.BoundPotential objects)."""

    N: int
    pressure: float
    temperature: float
    group_idxs: Any
    interval: int
    seed: int
    adaptive_scaling_enabled: bool = True
    initial_volume_scale_factor: Optional[float] = None

    def impl(self, bound_potentials):
        return custom_ops.MonteCarloBarostat(
            self.N,
            self.pressure,
            self.temperature,
            self.group_idxs,
            self.interval,
            bound_potentials,
            self.seed,
            self.adaptive_scaling_enabled,
            self.initial_volume_scale_factor or 0.0,  # 0.0 means "1 % of the initial box volume"
        )
