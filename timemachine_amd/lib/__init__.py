"""Thin dataclasses whose ``.impl()`` constructs ``custom_ops`` objects (reference: timemachine/lib/__init__.py:12-62)."""
from dataclasses import dataclass

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
