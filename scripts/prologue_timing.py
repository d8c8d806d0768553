import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from timemachine_amd import potentials as P, testsystems as ts
from timemachine_amd.lib import custom_ops as co
s = ts.dhfr_sized_water_box(); x = np.load("/tmp/ablate_frame.npy")
for prec in (np.float64, np.float32):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
    for _ in range(4): nb.execute(x, s.nb_params, s.box, True, False, False)
    buf, cnt = nb.debug_timing(8192); t = buf.reshape(-1)[:cnt].reshape(-1, 8)
    a = t[:, 0]
    f = [(a >> sh) & 0xffff for sh in (0, 16, 32, 48)]
    print(prec.__name__, "cycles to: first fetch issued %.0f | table copy issued %.0f | barrier passed %.0f | item loop entered %.0f ; total %.0f" % (f[0].mean(), f[1].mean(), f[2].mean(), f[3].mean(), t[:, 6].mean()))
