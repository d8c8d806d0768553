"""bench.py's config-3 CPU baseline alone (no GPU work): python scripts/cpu_baseline_only.py [old]
   old = the round-2/3 sampling (eight 384-row slabs, glibc's default mmap threshold), for comparison."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from timemachine_amd import testsystems as ts

s = ts.dhfr_shaped_box()
t0 = time.time()
if len(sys.argv) > 1 and sys.argv[1] == "old":
    bench._keep_host_heap = lambda: False
    r = bench.cpu_baseline(s, s.coords, 1.2, slabs_per_rep=8, rows_per_slab=384)
elif len(sys.argv) > 1 and sys.argv[1].isdigit():
    r = bench.cpu_baseline(s, s.coords, 1.2, rows_per_slab=int(sys.argv[1]))
else:
    r = bench.cpu_baseline(s, s.coords, 1.2)
print(sys.argv[1:] or "new", "threads", r["cores"], "estimates", [round(e, 2) for e in r["estimates_s"]], "spread", round(r["spread_rel"], 3), "wall", round(time.time() - t0, 1), flush=True)
