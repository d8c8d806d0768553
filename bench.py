#!/usr/bin/env python
"""Headline benchmark: ns/day of Langevin MD on a DHFR-sized (23 559-atom) explicit-water box at dt = 2.5 fs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f64|f32] [--cutoff 1.2]

Protocol = the reference's own (tests/test_benchmark.py:243-282): build the system, equilibrate untimed, then time
ctxt.multiple_steps(K) and report ns/day = K / t * 86400 * dt[ps] * 1e-3.  A "step" is one pass of the hot path: force
evaluation (NonbondedAllPairs + NonbondedExclusions + HarmonicBond + HarmonicAngle) + the BAOAB Langevin update, all
resident in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): every rank runs an independent replica
of the same workload (free-energy windows never interact during MD), `value` is the aggregate over ranks, scaling is
weak, and the only collective is the end-of-run gather of reduced potentials (RCCL over xGMI), outside the timed region
-- exactly where BASELINE's north star puts it.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       algorithmic HBM bytes of the dominant kernel (k_nonbonded_tiles) per launch / its measured duration
                 (HIP events on its launch stream) against the 8 TB/s HBM peak -- required form; the kernel is NOT
                 HBM-bound, so see roofline_valu for the roofline that actually binds it
  roofline_valu  algorithmic f64 flops per launch / duration against the 78.6 TFLOP/s FP64 vector peak
  cpu_baseline   the oracle's torch-f64 restatement of the reference's JAX path timed on the host cores on a bounded
                 sample of the same workload (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

DT = 2.5e-3  # ps
TEMPERATURE = 300.0
FRICTION = 1.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6  # = 1/2 of the 157.3 TFLOP/s FP32 vector peak (MI355X_MICROARCH.md chip table)
FP32_VALU_PEAK_TFLOPS = 157.3
C_PAIR_FLOPS = 280.0  # per interacting pair (rsqrt, erfc, exp, sincos, LJ, fixed-point conversions)
C_SLOT_FLOPS = 20.0  # per evaluated slot (min-image distance + cutoff test)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--cutoff", type=float, default=1.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=400)
    ap.add_argument("--windows", type=int, default=8, help="lambda windows of the end-of-run u_kl gather")
    ap.add_argument("--equil-scale", type=float, default=1.0, help="scale the untimed equilibration (profiling runs)")
    ap.add_argument("--equil-precision", choices=["f64", "f32"], default="f32")
    ap.add_argument("--parallel-children", action="store_true", help="SummedPotential(parallel=True) (accepted for interface parity; children always run in sequence)")
    ap.add_argument("--separate-potentials", action="store_true", help="pass the potentials to Context one by one (serial) instead of one SummedPotential")
    return ap.parse_args()


def init_distributed(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return rank, local_rank, world


def equilibrate(co, LangevinIntegrator, system, make_bps, seed, scale=1.0, prec=np.float32):
    """Lattice start -> liquid: short, strongly thermostatted stages with growing time step (untimed)."""
    x, v, box = system.coords.copy(), np.zeros_like(system.coords), system.box
    for dt, friction, steps in ((0.1e-3, 100.0, 600), (0.5e-3, 50.0, 600), (1.0e-3, 10.0, 800), (DT, FRICTION, 1000)):
        bps = make_bps(prec)
        ctxt = co.Context(x, v, box, LangevinIntegrator(TEMPERATURE, dt, friction, system.masses, seed).impl(), bps)
        ctxt.multiple_steps(max(int(steps * scale), 1), 0)
        x, v = ctxt.get_x_t(), ctxt.get_v_t()
        if not np.all(np.isfinite(x)):
            raise RuntimeError(f"equilibration diverged at dt={dt}")
        del ctxt
    return x, v


def count_pairs_within(x, box, cutoff):
    from scipy.spatial import cKDTree

    L = float(box[0, 0])
    xi = x - L * np.floor(x / L)
    xi = np.clip(xi, 0.0, np.nextafter(L, 0.0))
    tree = cKDTree(xi, boxsize=L)
    return int(tree.count_neighbors(tree, cutoff) - len(x)) // 2


def cpu_baseline(system, x, cutoff):
    """oracle (torch f64, row-blocked dense evaluation == the reference's JAX formulation) on the host cores,
    bounded sample: the first `rows` rows of the i<j pair matrix, scaled by the pair count."""
    import torch

    from oracle import ref_potentials as rp

    cores = min(os.cpu_count() or 1, 64)  # torch's elementwise kernels stop scaling long before 256 threads
    torch.set_num_threads(cores)
    N = system.num_atoms
    rows = 3072  # ~13 s of CPU work on 64 threads (the contract asks for a 10-30 s sample)
    xt = torch.tensor(x, requires_grad=True)
    pt = torch.tensor(system.nb_params)
    bt = torch.tensor(system.box)
    t0 = time.time()
    # evaluate only row blocks [0, rows): pairs (i, j > i) for i < rows
    total = 0.0
    for r0 in range(0, rows, 512):
        r1 = min(r0 + 512, rows)
        d3 = rp.delta_r(xt[r0:r1][:, None, :], xt[r0:][None, :, :], torch.diagonal(bt))
        d2 = (d3 * d3).sum(-1)
        upper = torch.arange(r0, r1)[:, None] < torch.arange(r0, N)[None, :]
        d2 = torch.where(upper, d2, torch.full_like(d2, 1e6))
        lj, es = rp._pair_energies(torch.sqrt(d2), pt[r0:r1, 0][:, None] * pt[None, r0:, 0], pt[r0:r1, 1][:, None] + pt[None, r0:, 1], pt[r0:r1, 2][:, None] * pt[None, r0:, 2], system.beta, cutoff)
        e = lj.sum() + es.sum()
        e.backward()
        total += float(e.detach())
    elapsed = time.time() - t0
    pairs_sample = sum(N - 1 - i for i in range(rows))
    pairs_full = N * (N - 1) // 2
    t_step = elapsed * pairs_full / pairs_sample
    return {
        "value": 86400.0 * DT * 1e-3 / t_step,
        "unit": "ns/day",
        "cores": cores,
        "kind": "port",
        "sample": f"du/dx of NonbondedAllPairs for rows 0..{rows - 1} of the i<j pair matrix ({pairs_sample / pairs_full:.1%} of all pairs, {elapsed:.1f} s), "
        f"scaled to the full matrix; torch f64 on {cores} threads; oracle restatement of the reference's dense JAX path, not JAX itself",
        "seconds_per_force_eval_extrapolated": t_step,
    }


def _walk(pot):
    yield pot
    if hasattr(pot, "get_potentials"):
        for c in pot.get_potentials():
            yield from _walk(c)


def find_all_pairs(bps):
    for bp in bps:
        for p in _walk(bp.get_potential()):
            if type(p).__name__.startswith("NonbondedAllPairs"):
                return p
    raise RuntimeError("no NonbondedAllPairs in the state")


def find_nonbonded(bps):
    """the FanoutSummedPotential([AllPairs, Exclusions]) that Nonbonded.to_gpu builds"""
    for bp in bps:
        for p in _walk(bp.get_potential()):
            if type(p).__name__ == "FanoutSummedPotential":
                return p
    raise RuntimeError("no Nonbonded in the state")


# k_nonbonded_tiles<Real, false, true, false>, per dispatch: (FETCH_SIZE + WRITE_SIZE) KB * 1024 -- profiles/r01_v9_pmc_f64.txt
PMC_TRAFFIC_BYTES = {"f64": (11652 + 44802) * 1024, "f32": (7215 + 33845) * 1024}


def main():
    args = parse_args()
    rank, local_rank, world = init_distributed(args.gpus)
    import torch

    from timemachine_amd import parallel
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    if co.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: timemachine_amd has no CPU fallback")
    co.set_device(local_rank)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    precision = np.float64 if args.precision == "f64" else np.float32
    system = ts.dhfr_sized_water_box(seed=2025, hmr=True, cutoff=args.cutoff)
    N = system.num_atoms

    def make_bps(prec):
        # one SummedPotential for the whole state -- how the reference packs a state
        # (fe/free_energy.py:614-657: make_summed_potential(...).to_gpu(np.float32) -> one BoundPotential)
        bps = ts.bound_potentials(system, prec)
        if args.separate_potentials:
            return [bp.to_gpu(prec).bound_impl for bp in bps]
        summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps], parallel=args.parallel_children)
        return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(prec).bound_impl]

    seed = 1234 + rank
    x, v = equilibrate(co, LangevinIntegrator, system, make_bps, seed, args.equil_scale, np.float64 if args.equil_precision == "f64" else np.float32)

    def run(prec, steps, warmup, profile_steps):
        bps = make_bps(prec)
        ctxt = co.Context(x, v, system.box, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, seed).impl(), bps)
        ctxt.multiple_steps(max(warmup, 1), 0)
        parallel.barrier()
        co.device_synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctxt.multiple_steps(steps, 0)
        co.device_synchronize()
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
        elapsed = parallel.max_over_ranks(elapsed)
        xf = ctxt.get_x_t()
        assert np.all(np.isfinite(xf)), "trajectory diverged"
        prof = None
        if profile_steps > 0:
            co.profile_reset()
            co.profile_set_enabled(True)
            ctxt.multiple_steps(profile_steps, 0)
            total_ms, launches = co.profile_read("nonbonded_tiles")
            co.profile_set_enabled(False)
            co.profile_reset()
            nb = find_all_pairs(bps)
            prof = {"kernel_ms": total_ms / max(launches, 1), "launches": launches, "tiles": nb.get_tile_ixn_count()}
        return elapsed, xf, ctxt, bps, prof

    elapsed, xf, ctxt, bps, prof = run(precision, args.steps, args.warmup, args.profile_steps if rank == 0 else 0)
    steps_per_s = args.steps / elapsed
    ns_day = steps_per_s * 86400.0 * DT * 1e-3 * world

    # ---- end-of-run reduced-potential gather (outside the timed region): each rank evaluates its final frame under
    # every window's parameters (charges scaled by lambda_l) and the rows are all-gathered over RCCL
    n_windows = max(args.windows, world)
    my_windows = parallel.windows_for_rank(world, world, rank)  # one replica per rank
    lambdas = np.linspace(0.0, 1.0, n_windows)
    params_l = np.stack([system.nb_params * np.array([1.0 - 0.1 * lam, 1.0, 1.0, 1.0]) for lam in lambdas])
    nb_pot = find_nonbonded(bps)
    _, _, u_row = nb_pot.execute_batch(xf[None], params_l, system.box[None], False, False, True)
    kT = 0.008314462618 * TEMPERATURE
    u_kl = parallel.gather_rows(my_windows, u_row.reshape(1, -1) / kT, world)
    gather_ok = bool(np.all(np.isfinite(u_kl)))

    if rank != 0:
        return

    out = {
        "metric": "ns/day (23k-atom solvated box, 2.5 fs) per GPU",
        "value": ns_day,
        "unit": "ns/day",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": "configs[2]: DHFR-sized explicit-water box, 7853 flexible TIP3P-like waters = 23559 atoms, box 6.223 nm, "
            f"direct-space erfc*switch electrostatics + LJ, cutoff {args.cutoff} nm, beta 2.0, HMR, dt 2.5 fs, Langevin 300 K friction 1/ps; "
            "one independent replica per GPU",
            "atoms": N,
            "potentials": f"HarmonicBond/HarmonicAngle/Nonbonded(AllPairs+Exclusions) *_{args.precision}, LangevinIntegrator<float>",
            "replicas": world,
        },
        "device": co.device_name(),
        "mbar_gather_ok": gather_ok,
    }

    if prof is not None:
        tiles = prof["tiles"]
        t_s = prof["kernel_ms"] * 1e-3
        p_int = count_pairs_within(xf, system.box, args.cutoff)
        rec = 64 if args.precision == "f64" else 32
        # algorithmic bytes per launch of the tile kernel: per atom one gathered record read + one
        # u64x3 force accumulator read-modify-write; per 32x32 tile 4 B tile id + 128 B of column indices
        bytes_alg = (rec + 48) * N + 132 * tiles
        flops_alg = C_PAIR_FLOPS * p_int + C_SLOT_FLOPS * 1024 * tiles
        peak_fl = FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else FP32_VALU_PEAK_TFLOPS
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "k_nonbonded_tiles",
            "achieved": bytes_alg / t_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": bytes_alg / t_s / 1e9 / HBM_PEAK_GBS,
            # FETCH_SIZE + WRITE_SIZE per dispatch of this kernel from the committed PMC passes (separate rocprofv3 --pmc runs
            # of this very command, scripts/gpu_pmc.sh -> profiles/r01_v9_pmc_f64.txt); counters uncalibrated for this
            # access pattern (MI355X_MICROARCH.md, HBM section).  ~20x the algorithmic bytes: the flush's u64 atomics.
            "traffic": PMC_TRAFFIC_BYTES.get(args.precision),
            "traffic_source": "profiles/r01_v9_pmc_f64.txt (FETCH_SIZE + WRITE_SIZE, KB per dispatch; measured in separate --pmc passes, not in this run)",
            "bytes_per_launch": bytes_alg,
            "kernel_ms": prof["kernel_ms"],
            "launches_timed": prof["launches"],
            "note": "required HBM form; this kernel is VALU-bound, not HBM-bound (SURVEY.md F10) -- see roofline_valu",
        }
        out["roofline_valu"] = {
            "bound": "valu_" + args.precision,
            "kernel": "k_nonbonded_tiles",
            "achieved": flops_alg / t_s / 1e12,
            "peak": peak_fl,
            "unit": "TFLOP/s",
            "frac": flops_alg / t_s / 1e12 / peak_fl,
            "flops_per_launch": flops_alg,
            "pairs_within_cutoff": p_int,
            "tiles_32x32": tiles,
            "tile_occupancy": p_int / (1024.0 * tiles) if tiles else None,
            "kernel_share_of_step": prof["kernel_ms"] / (1e3 * elapsed / args.steps),
        }

    if world == 1:
        # the other precision, for the record (the reference ships f32 kernels; BASELINE asks for f64 forces)
        other = np.float32 if precision == np.float64 else np.float64
        try:
            e2, _, _, _, _ = run(other, max(args.steps // 2, 1), max(args.warmup // 2, 1), 0)
            out["ns_day_" + ("f32" if other == np.float32 else "f64")] = (max(args.steps // 2, 1) / e2) * 86400.0 * DT * 1e-3
        except Exception as exc:  # pragma: no cover
            out["other_precision_error"] = str(exc)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system, xf, args.cutoff)

    print(json.dumps(out))


if __name__ == "__main__":
    main()
