"""GPU tests (``-m gpu``) of properties no single kernel owns: over thousands of steps of microcanonical MD the forces must
be the gradient of the energy the potentials report -- neighbor-list rebuilds, Hilbert re-sorts, the forces-only plan with its
hand-overs, the split tile kernels of small systems and the switched cutoff all sit between the two.  (The reference has no
such test; its pieces are tested one by one.  Size-independent, so it also runs at BASELINE config 3's full size.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOLTZ = 0.008314462618


@pytest.fixture(scope="module")
def co():
    from timemachine_amd.lib import custom_ops

    custom_ops.set_device(0)
    return custom_ops


def total_energy(bps, x, v, box, masses):
    u = sum(bp.execute(x, box, False, True)[1] for bp in bps)
    return u + 0.5 * float(np.sum(masses[:, None] * v * v)), u


@pytest.mark.parametrize(
    "system,precision,n_steps,tol",
    [
        ("config2", np.float64, 4000, 2.0e-3),
        ("config2", np.float32, 4000, 4.0e-3),
        ("config1", np.float64, 4000, 6.0e-3),
        ("config3", np.float64, 3000, 1.0e-3),
    ],
)
def test_microcanonical_energy_is_conserved(co, system, precision, n_steps, tol):
    """velocity Verlet at dt = 0.5 fs and at 0.25 fs over the same two picoseconds: the total energy neither drifts nor
    fluctuates by more than `tol` of the kinetic energy, and (f64) the fluctuation falls by the factor of four a second-order
    integrator owes -- which it would not if the forces were anything but the gradient of the reported energy"""
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, VelocityVerletIntegrator

    s = {"config1": lambda: ts.config1_water_cluster(3.0), "config2": ts.small_solvated_ligand, "config3": ts.dhfr_sized_water_box}[system]()
    masses = s.masses
    mk = lambda p: [bp.to_gpu(p).bound_impl for bp in ts.bound_potentials(s, p, nblist_padding=0.18 if system == "config3" else 0.1)]
    # thermalise (Langevin, f32 potentials), then switch the thermostat off
    eq = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(300.0, 1.0e-3, 5.0, masses, 3).impl(), mk(np.float32))
    eq.multiple_steps(3000, 0)
    x, v = eq.get_x_t(), eq.get_v_t()
    kinetic = 0.5 * float(np.sum(masses[:, None] * v * v))
    assert 0.8 < kinetic / (1.5 * s.num_atoms * BOLTZ * 300.0) < 1.2

    def nve(dt, steps, samples=16):
        bps = mk(precision)
        ctxt = co.Context(x, v, s.box, VelocityVerletIntegrator(dt, masses).impl(), bps)
        energies = []
        for _ in range(samples):
            e, _ = total_energy(bps, ctxt.get_x_t(), ctxt.get_v_t(), s.box, masses)
            assert np.isfinite(e)
            energies.append(e)
            ctxt.multiple_steps(steps // samples, 0)
        nb = bps[-1].get_potential().get_potentials()[0]
        return np.array(energies), nb.get_build_count()

    e1, builds = nve(0.5e-3, n_steps)
    e2, _ = nve(0.25e-3, 2 * n_steps)
    np.testing.assert_allclose(e1[0], e2[0], rtol=0, atol=1e-6 * kinetic)  # same start
    fluct1 = np.max(np.abs(e1 - e1[0])) / kinetic
    fluct2 = np.max(np.abs(e2 - e2[0])) / kinetic
    slope1 = np.polyfit(np.arange(len(e1)), e1, 1)[0] * len(e1) / kinetic
    assert fluct1 < tol and abs(slope1) < tol, (fluct1, fluct2, slope1)
    if precision == np.float64:
        rms1, rms2 = np.std(e1), np.std(e2)
        assert 2.0 < rms1 / rms2 < 8.0, (rms1, rms2, fluct1, fluct2)
    # ... and the runs went through list rebuilds, not one static list
    if system != "config1":
        assert builds > 5
