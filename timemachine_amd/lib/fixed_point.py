"""reference: timemachine/lib/fixed_point.py:8-15 (numpy instead of jax.numpy)."""
import numpy as np

from . import custom_ops


def fixed_to_float(v):
    """FIXED_TO_FLOAT, cpp/src/fixed_point.hpp:18-20."""
    return np.float64(np.int64(np.uint64(v))) / custom_ops.FIXED_EXPONENT


def float_to_fixed(v):
    """FLOAT_TO_FIXED as the reference's python mirror defines it (truncating cast)."""
    return np.uint64(np.int64(v * custom_ops.FIXED_EXPONENT))
