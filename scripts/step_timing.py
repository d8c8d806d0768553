#!/usr/bin/env python
"""How the per-step device time of Context.multiple_steps(K) depends on K (HIP events around the K steps) -- the
sensitivity of bench.py's protocol to --steps.  Prints per-call device / host times for runs of 20, 100, 500 steps."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    co.set_device(0)
    prec = np.float64
    system = ts.dhfr_sized_water_box(seed=2025, hmr=True)

    def make_bps(p):
        bps = ts.bound_potentials(system, p)
        summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
        return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]

    x, v = bench.equilibrate(co, LangevinIntegrator, system, make_bps, 1234)
    bps = make_bps(prec)
    ctxt = co.Context(x, v, system.box, LangevinIntegrator(300.0, bench.DT, 1.0, system.masses, 1234).impl(), bps)
    ctxt.multiple_steps(500, 0)
    nb = bench.find_all_pairs(bps)
    for K, reps in ((20, 12), (100, 6), (500, 3), (20, 6)):
        for _ in range(reps):
            b0 = nb.get_build_count()
            co.device_synchronize()
            t0 = time.perf_counter()
            ctxt.multiple_steps(K, 0)
            dev = ctxt.last_multiple_steps_ms()
            host = 1e3 * (time.perf_counter() - t0)
            print(f"K={K:4d} dev {1e3 * dev / K:7.2f} us/step  host {1e3 * host / K:7.2f} us/step  builds {nb.get_build_count() - b0}", flush=True)


if __name__ == "__main__":
    main()
