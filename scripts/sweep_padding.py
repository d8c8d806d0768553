#!/usr/bin/env python
"""nblist_padding sweep at BASELINE config 3 (23 559 atoms, rc 1.2): ns/day, rebuild period and list size per padding.

    python scripts/sweep_padding.py [--precision f64] [--paddings 0.05,0.1,0.15,0.2,0.3] [--steps 2000]

Results are padding-independent bitwise (tests/test_gpu_parity.py::test_bitwise_invariances); only the speed moves.
Prints one JSON line per point.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--paddings", default="0.05,0.1,0.15,0.2,0.3")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--settle", type=int, default=500)
    ap.add_argument("--cutoff", type=float, default=1.2)
    args = ap.parse_args()
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    co.set_device(0)
    prec = np.float64 if args.precision == "f64" else np.float32
    system = ts.dhfr_shaped_box(seed=2025, hmr=True, cutoff=args.cutoff)  # the bench workload

    def make_bps(p, padding=0.1):
        bps = ts.bound_potentials(system, p, nblist_padding=padding)
        summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
        return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(p).bound_impl]

    x, v = bench.equilibrate(co, LangevinIntegrator, system, make_bps, 1234)
    for pad in [float(t) for t in args.paddings.split(",")]:
        bps = make_bps(prec, pad)
        ctxt = co.Context(x, v, system.box, LangevinIntegrator(bench.TEMPERATURE, bench.DT, bench.FRICTION, system.masses, 1234).impl(), bps)
        ctxt.multiple_steps(args.settle, 0)
        nb = bench.find_all_pairs(bps)
        b0 = nb.get_build_count()
        co.device_synchronize()
        t0 = time.perf_counter()
        ctxt.multiple_steps(args.steps, 0)
        co.device_synchronize()
        el = time.perf_counter() - t0
        builds = nb.get_build_count() - b0
        co.profile_reset()
        co.profile_set_enabled(True)
        ctxt.multiple_steps(200, 0)
        total_ms, launches = co.profile_read("nonbonded_tiles")
        co.profile_set_enabled(False)
        co.profile_reset()
        tiles = nb.get_tile_ixn_count()
        for _ in range(8):  # reads 0 between the step that asked for a rebuild and the rebuild
            if tiles:
                break
            ctxt.multiple_steps(1, 0)
            tiles = nb.get_tile_ixn_count()
        print(json.dumps({
            "padding": pad, "precision": args.precision, "ns_day": args.steps / el * 86400 * bench.DT * 1e-3,
            "ms_per_step": 1e3 * el / args.steps, "steps_per_build": args.steps / max(builds, 1), "builds": builds,
            "tiles": tiles, "tile_kernel_us": 1e3 * total_ms / max(launches, 1),
        }), flush=True)
        del ctxt


if __name__ == "__main__":
    main()
