"""Energy-only execute_batch over F frames x P parameter sets: device time of a frame's first evaluation and of every further
parameter set on it, per potential of the RBFE state (all-atom Nonbonded on the DHFR-shaped box; Nonbonded / SummedPotential of
HostGuestSystem on the config-5-sized complex), identical parameter sets and five lambda windows.  GPU box only.
usage: python scripts/further_sets_probe.py [dhfr|config5] [f64|f32] [same|windows]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import custom_ops as co

co.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "dhfr"
prec = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
mode = sys.argv[3] if len(sys.argv) > 3 else "same"
F = 4
if which == "dhfr":
    s, n_lig = ts.dhfr_shaped_box(), 0
    pots = {"Nonbonded": (P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff), np.asarray(s.nb_params, dtype=np.float64).reshape(-1), None)}
else:
    s, n_lig = ts.config5_complex_sized(0.3), 40
    state = ts.rbfe_shaped_state(s, n_lig)
    sizes = [int(np.asarray(q).size) for _, q in state]
    flat = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])
    pots = {"Nonbonded": (P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff), np.asarray(s.nb_params, dtype=np.float64).reshape(-1), 0),
            "Summed(HostGuestSystem)": (P.SummedPotential([p for p, _ in state], [q for _, q in state]), flat, sum(sizes[:-1]))}
N = s.num_atoms
rng = np.random.default_rng(1)
xs = np.stack([s.coords + rng.normal(0, 0.002, s.coords.shape) for _ in range(F)])
boxes = np.stack([s.box] * F)

def sets(prm, n, lig_off):
    out = np.stack([prm] * n)
    if mode == "windows" and lig_off is not None and n_lig:
        for k in range(n):
            lam = 0.1 * k
            view = out[k][lig_off:].reshape(-1, 4)[N - n_lig:]
            view[:, 0] *= 1.0 - 0.5 * lam
            view[:, 3] = lam * s.cutoff
    return out

for label, (pot, prm, lig_off) in pots.items():
    impl = pot.to_gpu(prec).unbound_impl
    def dev_us(n_sets, reps=4):
        p = sets(prm, n_sets, lig_off)
        impl.execute_batch(xs, p, boxes, False, False, True)
        t = []
        for _ in range(reps):
            impl.execute_batch(xs, p, boxes, False, False, True)
            t.append(1e3 * co.debug_last_host_call_device_ms())
        return float(np.mean(t))
    one, five = dev_us(1), dev_us(5)
    print(f"{which} {prec.__name__} {mode} {label}: {F} frames x 1 set {one / F:.1f} us per frame (first evaluation); x 5 sets {five / (5 * F):.1f} us per execution"
          f" -> every further parameter set {(five - one) / (4 * F):.1f} us", flush=True)
