"""The long form of scripts/fuzz_campaign.py: <count> seeds from <first seed> on, 16 runs per seed (single windows, windows sharing the GPU,
the benchmark composition; both precisions; listed and static lists).   python scripts/fuzz_campaign_long.py <count> <first seed>   (GPU)"""
import sys, numpy as np, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_interleavings as T
from timemachine_amd.lib import custom_ops as co
from timemachine_amd import potentials as P
co.set_device(0)
t0 = time.time(); nbad = 0; n = 0
def cmp(tag, fast, plain, fin_slice):
    global nbad, n
    n += 1
    bad = [k for k, (a, b) in enumerate(zip(fast, plain)) if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)]
    fin = all(np.all(np.isfinite(a)) for a in fast[fin_slice])
    if bad or not fin or len(fast) != len(plain):
        nbad += 1
        print(tag, "MISMATCH" if bad else "", "NONFINITE" if not fin else "", bad[:4], len(fast), len(plain), flush=True)
for seed in range(int(sys.argv[2]), int(sys.argv[2]) + int(sys.argv[1])):
    for which, sk, nops in (("config2", 0, 90), ("config2", 4608, 90), ("config4", 0, 50)):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed, nops)
            f, _, _ = T._run(co, P, which, prec, sk, True, ops); p_, _, _ = T._run(co, P, which, prec, sk, False, ops)
            cmp(f"single-window seed {seed} {which} {sk} {prec.__name__}", f, p_, slice(-5, -2))
    for sk in (0, 4608):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed + 7, 40)
            f, _ = T._run_windows(co, P, prec, sk, True, ops); p_, _ = T._run_windows(co, P, prec, sk, False, ops)
            cmp(f"windows seed {seed} {sk} {prec.__name__}", f, p_, slice(-9, None))
    for sk, packed in ((0, False), (4608, False), (0, True)):
        for prec in (np.float64, np.float32):
            ops = T._make_ops(seed + 13, 60)
            f, _ = T._run_single(co, prec, sk, False, ops, packed); p_, _ = T._run_single(co, prec, sk, True, ops, packed)
            cmp(f"bench-composition seed {seed} {sk} {packed} {prec.__name__}", f, p_, slice(-5, -2))
    if (seed % 50) == 0: print("...", seed, n, nbad, f"{time.time()-t0:.0f}s", flush=True)
print(f"big campaign: {nbad} bad of {n} in {time.time()-t0:.0f} s")
