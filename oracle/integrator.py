"""BAOAB Langevin step model.  TEST INFRASTRUCTURE ONLY.

``langevin_coefficients`` / ``baoab_step``: timemachine/integrator.py:15-53,124-150 (f64, python BOLTZ).
``baoab_step_device_model``: the mixed-precision arithmetic of cpp/src/kernels/k_integrator.cuh:5-62 with
the coefficients of cpp/src/langevin_integrator.cu:14-33 (Real = float in the bound class,
cpp/src/wrap_kernels.cpp:700; BOLTZ from cpp/src/constants.hpp:5).
"""
import numpy as np

BOLTZ_PY = 1.380658e-23 * 6.0221367e23 / 1000  # timemachine/constants.py:5-8
BOLTZ_CPP = 0.008314462618  # cpp/src/constants.hpp:5


def langevin_coefficients(temperature, dt, friction, masses, boltz=BOLTZ_PY):
    kT = boltz * temperature
    nscale = np.sqrt(kT / masses)
    ca = np.exp(-friction * dt)
    cb = dt / masses
    cc = np.sqrt(1 - np.exp(-2 * friction * dt)) * nscale
    return ca, cb, cc


def baoab_step(x, v, force, noise, ca, cb, cc, dt):
    """integrator.py:137-144: v_mid = v + cb f; v' = ca v_mid + cc n; x' = x + dt/2 (v_mid + v')."""
    v_mid = v + cb[:, None] * force
    new_v = ca * v_mid + cc[:, None] * noise
    new_x = x + 0.5 * dt * (v_mid + new_v)
    return new_x, new_v


def device_coefficients(temperature, dt, friction, masses, real=np.float32):
    """langevin_integrator.cu:17-33: computed in f64 (dt already rounded to Real), stored as Real."""
    dt_r = real(dt)
    ca = real(np.exp(-friction * dt))
    kT = BOLTZ_CPP * temperature
    adj = np.sqrt(1 - np.exp(-2 * friction * dt))
    cb = (np.float64(dt_r) / masses).astype(real)
    cc = (adj * np.sqrt(kT / masses)).astype(real)
    return ca, cb, cc, dt_r


def baoab_step_device_model(x, v, du_dx_fixed, noise, ca, cb, cc, dt_r, real=np.float32):
    """k_integrator.cuh:28-46 with x, v stored f64 and the arithmetic in ``real``."""
    from .fixed_point import fixed_to_float

    r = real
    force = -(fixed_to_float(du_dx_fixed).astype(r))
    # RealType v_mid = v_t[...] (double) + cbs * force (Real * Real): the sum is formed in double, then rounded to Real
    v_mid = (v + (cb[:, None] * force).astype(np.float64)).astype(r)
    new_v_r = (ca * v_mid + cc[:, None] * noise.astype(r)).astype(r)
    new_v = new_v_r.astype(np.float64)
    # x += 0.5*dt*(v_mid + v_t): RealType * (RealType + double) -> evaluated in double
    new_x = x + np.float64(r(0.5) * dt_r) * (v_mid.astype(np.float64) + new_v)
    return new_x, new_v


def velocity_verlet_device_model(x, v, force_fixed_fn, cbs, dt, n_steps):
    """The reference's device Velocity Verlet (verlet_integrator.cu:22-111, k_integrator.cuh:64-130) in double:
    initialize (half kick + drift), n_steps - 1 full (kick + drift) steps, finalize (half kick).  ``force_fixed_fn(x)``
    returns the uint64 fixed-point du/dx; ``cbs`` = -dt / mass.  Equivalent schedule: integrator.py:169-199."""
    from .fixed_point import fixed_to_float

    x, v = x.copy(), v.copy()
    v = v + (0.5 * cbs)[:, None] * fixed_to_float(force_fixed_fn(x))
    x = x + dt * v
    for _ in range(n_steps - 1):
        v = v + cbs[:, None] * fixed_to_float(force_fixed_fn(x))
        x = x + dt * v
    v = v + (0.5 * cbs)[:, None] * fixed_to_float(force_fixed_fn(x))
    return x, v
