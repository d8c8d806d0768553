"""times execute_batch (frames x 5 parameter sets) of every term of the RBFE state on its own and of the SummedPotential, per output form
usage: python scripts/potentials_probe.py [config5|config4]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import custom_ops as co

co.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "config5"
s, n_lig = (ts.config5_complex_sized(0.3), 40) if which == "config5" else (ts.config4_solvated_ligand(0.3), 30)
state = ts.rbfe_shaped_state(s, n_lig)
rng = np.random.default_rng(1)
xs = np.stack([s.coords + rng.normal(0, 0.002, s.coords.shape) for _ in range(4)])
boxes = np.stack([s.box] * 4)
forms = {"all": (True, True, True), "dx": (True, False, False), "dp": (False, True, False), "u": (False, False, True)}
def t(impl, prm):
    prm = np.stack([np.asarray(prm, dtype=np.float64).reshape(-1)] * 5)
    out = {}
    for name, fl in forms.items():
        impl.execute_batch(xs, prm, boxes, *fl)
        co.device_synchronize(); t0 = time.perf_counter()
        impl.execute_batch(xs, prm, boxes, *fl)
        out[name] = 1e6 * (time.perf_counter() - t0) / 20
    return out
for prec in (np.float32, np.float64):
    for pot, prm in state:
        print(prec.__name__, type(pot).__name__.ljust(30), np.asarray(prm).size, {k: round(v, 1) for k, v in t(pot.to_gpu(prec).unbound_impl, prm).items()})
    summed = P.SummedPotential([p for p, _ in state], [q for _, q in state]).to_gpu(prec).unbound_impl
    print(prec.__name__, "Summed".ljust(30), {k: round(v, 1) for k, v in t(summed, np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state])).items()})
