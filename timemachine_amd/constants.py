"""Physical constants of the reference (timemachine/constants.py:5-16,52-60)."""
from enum import IntEnum

BOLTZMANN = 1.380658e-23  # J/kelvin
AVOGADRO = 6.0221367e23  # mol^-1
RGAS = BOLTZMANN * AVOGADRO  # J/mol per kelvin
BOLTZ = RGAS / 1000  # kJ/mol per kelvin
ONE_4PI_EPS0 = 138.935456
DEFAULT_TEMP = 300.0  # kelvin
DEFAULT_PRESSURE = 1.013  # bar
DEFAULT_KT = BOLTZ * DEFAULT_TEMP


class NBParamIdx(IntEnum):
    Q_IDX = 0
    LJ_SIG_IDX = 1
    LJ_EPS_IDX = 2
    W_IDX = 3
