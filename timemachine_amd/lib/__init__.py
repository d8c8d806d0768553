"""Picklable descriptions of the integrators / movers of an MD run; ``.impl()`` builds the ``custom_ops`` object.

The SCHEMA is the reference's (timemachine/lib/__init__.py:12-62) because callers construct these positionally and
by keyword: class names, field names, field order, defaults, and ``impl()`` / ``impl(bound_potentials)``.  Which
``custom_ops`` constructor argument each field feeds is stated once, in ``_CTOR_ARGS``, next to the wrap_kernels.cpp lines
that define that constructor.
"""
import os
import sys
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np
from numpy.typing import NDArray

# `custom_ops` is the compiled pybind11 module (csrc/wrap_custom_ops.cpp -> lib/custom_ops.<abi>.so), like the reference's.
# TM_AMD_BINDING=ctypes swaps in the ctypes mirror of the same surface (custom_ops_ctypes.py) -- both sit on the same C ABI and
# the test suite runs against both; TM_AMD_LIB (a variant build of the library, A/B measurements) implies it, because the
# compiled module is linked against the product library.
if os.environ.get("TM_AMD_BINDING", "").lower() == "ctypes" or os.environ.get("TM_AMD_LIB"):
    from . import custom_ops_ctypes as custom_ops

    sys.modules[__name__ + ".custom_ops"] = custom_ops
else:
    try:
        from . import custom_ops
    except ImportError as e:  # no CPU fallback, no silent second choice: say what is missing
        raise ImportError(
            f"timemachine_amd.lib.custom_ops (the compiled pybind11 module) could not be imported: {e}. Build it first: "
            "python -c 'import __graft_entry__ as g; g.build()' or python -m timemachine_amd.csrc.build"
        ) from e

# custom_ops constructor argument order, by class (field names of the dataclass; "*" = the objects handed to impl())
_CTOR_ARGS = {
    "LangevinIntegrator": ("masses", "temperature", "dt", "friction", "seed"),  # wrap_kernels.cpp:699-715
    "VelocityVerletIntegrator": ("dt", "cbs"),  # wrap_kernels.cpp:717-729
    "MonteCarloBarostat": (  # wrap_kernels.cpp:196-294
        "N", "pressure", "temperature", "group_idxs", "interval", "*", "seed", "adaptive_scaling_enabled", "initial_volume_scale_factor",
    ),
}


class _Described:
    """impl(): look the constructor up by the dataclass' own name and feed it the fields in _CTOR_ARGS order."""

    def _ctor_value(self, name):
        return getattr(self, name)

    def impl(self, *handed_over):
        cls_name = type(self).__name__
        args = [handed_over[0] if name == "*" else self._ctor_value(name) for name in _CTOR_ARGS[cls_name]]
        return getattr(custom_ops, cls_name)(*args)


@dataclass
class LangevinIntegrator(_Described):
    temperature: float
    dt: float
    friction: float
    masses: NDArray[np.float64]
    seed: int


@dataclass
class VelocityVerletIntegrator(_Described):
    dt: float
    masses: NDArray[np.float64]

    cbs: NDArray[np.float64] = field(init=False)  # what the device kernel multiplies du/dx with: -dt / m

    def __post_init__(self):
        self.cbs = np.negative(self.dt / np.asarray(self.masses, dtype=np.float64))


@dataclass
class MonteCarloBarostat(_Described):
    N: int
    pressure: float
    temperature: float
    group_idxs: Any
    interval: int
    seed: int
    adaptive_scaling_enabled: bool = True
    initial_volume_scale_factor: Optional[float] = None

    def _ctor_value(self, name):
        if name == "initial_volume_scale_factor":
            # None (and 0) reach the device as 0.0, its "start at 1 % of the box volume" marker (barostat.cu:167-171)
            return float(self.initial_volume_scale_factor) if self.initial_volume_scale_factor else 0.0
        return getattr(self, name)

    def impl(self, bound_potentials):
        return super().impl(bound_potentials)
