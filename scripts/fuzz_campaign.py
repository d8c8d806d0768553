"""A longer campaign of tests/test_gpu_interleavings.py: random interleavings of stepping / evaluations / setters / batches with every
fast path on against every fast path off, over many seeds (the test suite keeps a handful).  Prints the seeds that disagree.
    python scripts/fuzz_campaign.py      (GPU)"""
import sys, numpy as np, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_interleavings as T
from timemachine_amd.lib import custom_ops as co
from timemachine_amd import potentials as P
co.set_device(0)
t0 = time.time()
nbad = 0
for seed in range(100, 160):
    for which, sk, n in (("config2", 0, 80), ("config2", 4608, 80), ("config4", 0, 40)):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed, n)
            fast, pf, labels = T._run(co, P, which, prec, sk, True, ops)
            plain, pp, _ = T._run(co, P, which, prec, sk, False, ops)
            bad = [k for k, (a, b) in enumerate(zip(fast, plain)) if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)]
            fin = all(np.all(np.isfinite(a)) for a in fast[-5:-2])
            if bad or not fin:
                nbad += 1
                print("seed", seed, which, sk, prec.__name__, "MISMATCH" if bad else "", "NONFINITE" if not fin else "", [(k, labels[k]) for k in bad[:3]], flush=True)
print("campaign done", nbad, "bad of", 60 * 3 * 2, f"{time.time()-t0:.0f}s")
nbad = 0
for seed in range(200, 230):
    for sk in (0, 4608):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed, 40)
            fast, sf = T._run_windows(co, P, prec, sk, True, ops)
            plain, sp = T._run_windows(co, P, prec, sk, False, ops)
            bad = [k for k, (a, b) in enumerate(zip(fast, plain)) if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)]
            fin = all(np.all(np.isfinite(a)) for a in fast[-9:])
            if bad or not fin:
                nbad += 1
                print("windows seed", seed, sk, prec.__name__, "MISMATCH" if bad else "", "NONFINITE" if not fin else "", bad[:4], len(fast), flush=True)
print("windows campaign done", nbad, "bad of", 30 * 2 * 2)
nbad = 0
for seed in range(300, 340):
    for sk, packed in ((0, False), (4608, False), (0, True)):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed, 60)
            fast, pf = T._run_single(co, prec, sk, False, ops, packed)
            plain, pp = T._run_single(co, prec, sk, True, ops, packed)
            bad = [k for k, (a, b) in enumerate(zip(fast, plain)) if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)]
            fin = all(np.all(np.isfinite(a)) for a in fast[-5:-2])
            if bad or not fin:
                nbad += 1
                print("single seed", seed, sk, packed, prec.__name__, "MISMATCH" if bad else "", "NONFINITE" if not fin else "", bad[:4], len(fast), flush=True)
print("single-Nonbonded campaign done", nbad, "bad of", 40 * 3 * 2)
