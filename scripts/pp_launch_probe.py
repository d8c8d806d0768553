"""The tile kernel's launch time per output form on the DHFR-shaped box (23 559 atoms): execute_batch over 4 frames x 5 parameter sets of
the all-atom NonbondedAllPairs in every form, nothing else in the process -- run under rocprofv3 --kernel-trace --stats
(scripts/gpu_stats_cmd.sh) to read the forces-only, energy-only, u + du/dx, du/dx + du/dp and u + du/dx + du/dp instantiations' averages.
GPU box only.  usage: python scripts/pp_launch_probe.py [f64|f32]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import custom_ops as co

co.set_device(0)
co.debug_set_energy_memo(False)  # (every energy-only evaluation launches its tiles)
prec = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
s = ts.dhfr_shaped_box()
rng = np.random.default_rng(1)
xs = np.stack([s.coords + rng.normal(0, 0.002, s.coords.shape) for _ in range(4)])
boxes = np.stack([s.box] * 4)
prm = np.stack([np.asarray(s.nb_params, dtype=np.float64)] * 5)
impl = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff, nblist_padding=0.1).to_gpu(prec).unbound_impl
for flags in ((True, False, False), (False, False, True), (True, False, True), (True, True, False), (True, True, True), (False, True, False)):
    for _ in range(3):
        impl.execute_batch(xs, prm, boxes, *flags)
    print(flags, "device us per execution %.1f" % (1e3 * co.debug_last_host_call_device_ms() / 20), flush=True)
