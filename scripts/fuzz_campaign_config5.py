"""tests/test_gpu_interleavings.py's single-window runs at config-5 size (31k atoms, 40-atom ligand: two guest row blocks), every fast
path on against every fast path off, over <count> seeds from <first seed> on (the size at which the carrier take-over defect showed).
    python scripts/fuzz_campaign_config5.py <count> <first seed>   (GPU)"""
import sys, numpy as np, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_interleavings as T
from timemachine_amd.lib import custom_ops as co
from timemachine_amd import potentials as P
co.set_device(0)
t0 = time.time(); nbad = 0; n = 0
for seed in range(int(sys.argv[2]), int(sys.argv[2]) + int(sys.argv[1])):
    for prec in (np.float64, np.float32):
        ops = T._make_ops(seed, 40)
        fast, _, labels = T._run(co, P, "config5", prec, 0, True, ops)
        plain, _, _ = T._run(co, P, "config5", prec, 0, False, ops)
        bad = [k for k, (a, b) in enumerate(zip(fast, plain)) if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)]
        fin = all(np.all(np.isfinite(a)) for a in fast[-5:-2])
        n += 1
        if bad or not fin or len(fast) != len(plain):
            nbad += 1
            print("seed", seed, prec.__name__, "MISMATCH" if bad else "", "NONFINITE" if not fin else "", [(k, labels[k]) for k in bad[:3]], flush=True)
    if time.time() - t0 > float(sys.argv[3]) if len(sys.argv) > 3 else False:
        break
print(f"config-5 campaign: {nbad} bad of {n} in {time.time()-t0:.0f} s")
