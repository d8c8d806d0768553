"""MD steps of one window with the reference's RBFE state composition (testsystems.rbfe_shaped_state) -- what rocprofv3 traces for
profiles/r06_*_per_step_rbfe_*.txt (scripts/gpu_profile_rbfe.sh).
usage: python scripts/rbfe_steps.py <config4|config5> <f64|f32> [--no-merge] [--single] [--barostat N] [--steps K] [--padding P]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (equilibrate, constants)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("size", choices=["config2", "config4", "config5"])
    ap.add_argument("precision", choices=["f64", "f32"])
    ap.add_argument("--no-merge", action="store_true", help="the two tile producers each for itself (rounds 1-5)")
    ap.add_argument("--single", action="store_true", help="one all-atom Nonbonded instead of the composition (the benchmark states' shape)")
    ap.add_argument("--barostat", type=int, default=0)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--padding", type=float, default=0.18)
    ap.add_argument("--equil-scale", type=float, default=0.5)
    args = ap.parse_args()
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, custom_ops as co

    co.set_device(0)
    system, n_lig = {"config2": (ts.small_solvated_ligand(0.3), 20), "config4": (ts.config4_solvated_ligand(0.3), 30), "config5": (ts.config5_complex_sized(0.3), 40)}[args.size]
    prec = np.float64 if args.precision == "f64" else np.float32
    co.debug_set_merge_producers(not args.no_merge)

    def make_bps(p, single=args.single):
        bound = ts.bound_potentials(system, p, nblist_padding=args.padding) if single else ts.rbfe_bound_potentials(system, n_lig, nblist_padding=args.padding)
        summed = P.SummedPotential([bp.potential for bp in bound], [bp.params for bp in bound])
        return [summed.bind_params_list([bp.params for bp in bound]).to_gpu(p).bound_impl]

    x, v = bench.equilibrate(co, LangevinIntegrator, system, make_bps, 7, args.equil_scale, np.float32)
    bps = make_bps(prec)
    movers = [MonteCarloBarostat(system.num_atoms, 1.0, bench.TEMPERATURE, ts.molecule_groups(system), args.barostat, 3).impl(bps)] if args.barostat else []
    ctxt = co.Context(x, v, system.box, LangevinIntegrator(bench.TEMPERATURE, bench.DT, bench.FRICTION, system.masses, 5).impl(), bps, movers=movers)
    ctxt.multiple_steps(bench.SETTLE_STEPS, 0)
    ctxt.multiple_steps(args.steps, 0)
    ms = ctxt.last_multiple_steps_ms()
    nb = bench.find_all_pairs(bps)
    rec = {"size": args.size, "atoms": system.num_atoms, "precision": args.precision, "merged": not args.no_merge and not args.single, "single_nonbonded": args.single,
           "us_per_step": 1e3 * ms / args.steps, "ns_day": args.steps / (1e-3 * ms) * 86400.0 * bench.DT * 1e-3, "merged_stats": nb.get_merged_stats()}
    if movers:
        rec["barostat_attempt_paths"] = movers[0].get_attempt_paths()
    print(rec)


if __name__ == "__main__":
    main()
