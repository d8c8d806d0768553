"""tests/test_gpu_random_parity.py over many more seeds: random small periodic systems (sizes, boxes, cutoffs, clashes, 4D offsets,
exclusions, subsets, groups) on the GPU against the CPU oracle.   python scripts/fuzz_parity.py [first_seed] [count]   (GPU)"""
import sys, time, traceback
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_random_parity as T
from timemachine_amd.lib import custom_ops as co
co.set_device(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    for prec in (np.float64, np.float32):
        try:
            T.run_case(seed, prec)
            T.run_bonded_case(seed, prec)
        except AssertionError as e:
            bad += 1
            print("FAIL", seed, prec.__name__, str(e)[:300], flush=True)
        except Exception:
            bad += 1
            print("ERROR", seed, prec.__name__, traceback.format_exc()[-400:], flush=True)
print(f"parity campaign: {bad} failures of {2 * count} cases in {time.time() - t0:.0f} s")
