"""Potential / BoundPotential dataclass plumbing (reference: timemachine/potentials/potential.py:20-110).

``Potential.to_gpu(precision)`` looks up ``custom_ops.<ClassName>_<f32|f64>`` and calls it with ``astuple(self)`` --
dataclass field order == constructor argument order -- exactly as the reference does (potential.py:28-37).

The reference's ``__call__`` is its JAX (CPU) implementation; this package has no CPU implementation of the physics
on the product side (the CPU restatement lives under oracle/ and is test infrastructure only), so ``__call__`` evaluates
the energy with the f64 HIP kernels.
"""
from abc import ABC
from dataclasses import astuple, dataclass
from typing import Any, Generic, TypeVar

import numpy as np
from numpy.typing import NDArray

from ..lib import custom_ops

Precision = Any
_P = TypeVar("_P", bound="Potential", covariant=True)


def get_custom_ops_class_name_suffix(precision: Precision) -> str:
    # reference: potential.py:74-80
    if precision == np.float32:
        return "f32"
    elif precision == np.float64:
        return "f64"
    else:
        raise ValueError("invalid precision")


@dataclass
class Potential(ABC):
    def __call__(self, conf, params, box) -> float:
        return self.to_gpu(np.float64)(np.asarray(conf), np.asarray(params), _box_or_vacuum(box))

    def bind(self: _P, params) -> "BoundPotential[_P]":
        return BoundPotential(self, params)

    def to_gpu(self, precision: Precision) -> "GpuImplWrapper":
        ctor = getattr(custom_ops, self._custom_ops_class_name(precision))
        args = astuple(self)
        impl = ctor(*args)
        return GpuImplWrapper(impl)

    @classmethod
    def _custom_ops_class_name(cls, precision: Precision) -> str:
        suffix = get_custom_ops_class_name_suffix(precision)
        return f"{cls.__name__}_{suffix}"


def _box_or_vacuum(box):
    # bonded terms ignore the box; the reference's tests pass a huge box for "vacuum" (tests/test_bonded.py:26)
    return np.eye(3) * 100.0 if box is None else np.asarray(box)


@dataclass
class BoundPotential(Generic[_P]):
    potential: _P
    params: Any

    def __call__(self, conf, box) -> float:
        return self.potential(conf, self.params, box)

    def to_gpu(self, precision: Precision) -> "BoundGpuImplWrapper":
        return self.potential.to_gpu(precision).bind(np.asarray(self.params))


@dataclass
class GpuImplWrapper:
    unbound_impl: custom_ops.Potential

    def __call__(self, conf: NDArray, params: NDArray, box: NDArray) -> float:
        # reference: jax_interface.call_unbound_impl (potentials/jax_interface.py:12-40): energy of execute()
        _, _, u = self.unbound_impl.execute(conf, params, box, False, False, True)
        return u

    def bind(self, params: NDArray) -> "BoundGpuImplWrapper":
        return BoundGpuImplWrapper(custom_ops.BoundPotential(self.unbound_impl, params))


@dataclass
class BoundGpuImplWrapper:
    bound_impl: custom_ops.BoundPotential

    def __call__(self, conf: NDArray, box: NDArray) -> float:
        _, u = self.bound_impl.execute(conf, box, False, True)
        return u


def _first_of_type(items, kind, of=lambda item: item):
    hits = (item for item in items if isinstance(of(item), kind))
    try:
        return next(hits)
    except StopIteration:
        raise ValueError(f"Unable to find potential of type: {kind}") from None


def get_bound_potential_by_type(bps, pot_type):
    """The first bound potential of `bps` whose potential is a `pot_type`; ValueError when there is none.
    reference: timemachine/potentials/potential.py:82-98 (callers: md/minimizer.py, md/enhanced.py, md/barostat/moves.py,
    fe/free_energy.py, fe/absolute_hydration.py)."""
    return _first_of_type(bps, pot_type, of=lambda bp: bp.potential)


def get_potential_by_type(pots, pot_type):
    """The first potential of `pots` that is a `pot_type`; ValueError when there is none.
    reference: timemachine/potentials/potential.py:101-116."""
    return _first_of_type(pots, pot_type)
