"""Per-workgroup phase timings of k_find_ixns (needs the -DTM_NBL_TIMING variant: build.build_variant("nbltiming", ["TM_NBL_TIMING"]);
run with TM_AMD_LIB pointing at it).  Prints device printf lines: rows / coarse / pass2 / cost / publish in 10 ns ticks."""
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from timemachine_amd import potentials as P, testsystems as ts
s = ts.dhfr_sized_water_box()
for prec in (np.float64, np.float32):
    nb = P.NonbondedAllPairs(s.num_atoms, s.beta, s.cutoff).to_gpu(prec).unbound_impl
    nb.execute(s.coords, s.nb_params, s.box, True, False, False)
    print("----", prec.__name__, flush=True)
