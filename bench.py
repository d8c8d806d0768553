#!/usr/bin/env python
"""Headline benchmark: ns/day of Langevin MD on a DHFR-sized (23 559-atom) explicit-water box at dt = 2.5 fs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f64|f32] [--cutoff 1.2] [--mode md|hrex]

Protocol = the reference's own (tests/test_benchmark.py:243-282): build the system, equilibrate untimed, then time
ctxt.multiple_steps(K) and report ns/day = K / t * 86400 * dt[ps] * 1e-3.  A "step" is one pass of the hot path: force
evaluation (NonbondedAllPairs + NonbondedExclusions + HarmonicBond + HarmonicAngle + PeriodicTorsion) + the BAOAB Langevin update, all
resident in HBM.  The default workload carries every term the north star names: 7 023 waters + a 2 490-atom solute with
bonds, angles, proper + improper-form PeriodicTorsion terms and 1-4 exclusions at partial scales (DHFR's shape,
testsystems/dhfr.py:9-23); `--workload water` is the pure-water box of rounds 1 and 2.  The record also carries the same box
at the reference's own protocol (cutoff 1.0 nm, f32: `rc1.0_f32`; and `rc1.0_f64`).

Timing.  The timed Context is settled with SETTLE_STEPS untimed steps (part of the equilibration: first list builds,
allocator and clock ramp), then W warm-up steps, then EXACTLY K steps between barrier + device synchronisation on both
sides.  `value` comes from HIP events recorded on the Context's stream right before the first and right after the last
of those K steps (tm_context_last_multiple_steps_ms), maximum over ranks; the host wall clock around the same call is
reported next to it (`host_ms_per_step`: it additionally contains the final frame's device-to-host copy and the Python
call overhead, 0.3 ms in total -- visible at K = 20, invisible at K = 2000).

N > 1.  `python bench.py --gpus N` launches N ranks itself (torch.distributed.run, one per GPU, RCCL); under an
existing launcher (RANK / WORLD_SIZE set) it joins that job and insists that WORLD_SIZE == --gpus.
  --mode md    every rank runs an independent replica of the same workload (free-energy windows never interact during
               MD); `value` is the aggregate over ranks, scaling is weak, and the only collective is the end-of-run
               gather of reduced potentials (RCCL over xGMI), outside the timed region -- where BASELINE's north star
               puts it.  Its latency is reported.
  --mode hrex  BASELINE config 5's shape: 24 lambda windows of a ~31k-atom state dealt round-robin to the ranks
               (parallel.windows_for_rank), 400 MD steps per frame, then one exchange step: the (replica, state) energy
               matrix rows of the resident replicas (execute_batch_sparse, max_delta_states = 4), ONE all_gather, the
               identical seeded swap chain on every rank, parameters re-bound (fe/free_energy.py:1148-1200,1537-1551).
               Reports aggregate ns/day, frames/s and the measured exchange latency.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       algorithmic HBM bytes of the dominant kernel (k_nonbonded_tiles) per launch / its measured duration
                 (HIP events on its launch stream) against the 8 TB/s HBM peak -- required form; the kernel is NOT
                 HBM-bound, so see roofline_valu for the roofline that actually binds it
  roofline_valu  algorithmic f64 flops per launch / duration against the 78.6 TFLOP/s FP64 vector peak
  cpu_baseline   the oracle's torch-f64 restatement of the reference's JAX path timed on the host cores: one whole force
                 evaluation of the same workload per pass, 5 timed passes (rank 0, N == 1 only); cpu_baseline_configs: BASELINE
                 configs 1 (500 BAOAB steps in each box) and 2 (u + du_dx + du_dp, 10 repetitions) the same way
"""
import argparse
import dataclasses
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

DT = 2.5e-3  # ps
TEMPERATURE = 300.0
FRICTION = 1.0
SETTLE_STEPS = 500  # untimed, on the timed Context, whatever --warmup says
SECONDARY_STEPS = 500  # timed steps of the legs that are not `value` (other precision, cutoff 1.0, NPT)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6  # = 1/2 of the 157.3 TFLOP/s FP32 vector peak (MI355X_MICROARCH.md chip table)
FP32_VALU_PEAK_TFLOPS = 157.3
C_PAIR_FLOPS = 280.0  # per interacting pair (rsqrt, erfc, exp, sincos, LJ, fixed-point conversions)
C_SLOT_FLOPS = 20.0  # per evaluated slot (min-image distance + cutoff test)
# The same two figures counted on THIS kernel's ISA (f64 forces-only batch body on a flat item / Gram-form filter round, DESIGN.md
# section 4.2; an FMA = 2): displacement 3, d^2 5, table index + degree-5 Horner 11, charges 2, LJ 20, prefactor 1, fixed-point
# products + magic adds 6 = 48 per interacting pair; filter: 3 FMA = 6 per evaluated slot (the row's |r|^2 arrives by a move, the
# threshold carries the column's |c|^2).  The table-driven kernel simply executes fewer flops than the reference formula SURVEY priced.
OWN_PAIR_FLOPS = {"f64": 48.0, "f32": 75.0}  # f32: analytic erfc / exp / switch instead of the table (rsq, rcp, exp, sin, cos count 1)
OWN_SLOT_FLOPS = 6.0
CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md: peak engine clock; used only to turn a kernel time into cycles for valu_busy
LDS_COUNTER_SATURATED = 1.9  # SQ_ACTIVE_INST_LDS * 4 / (CUs * cycles) of a kernel that keeps every CU's LDS pipe busy (measured: 1.84-1.98)
METRIC = "ns/day (23k-atom solvated box, 2.5 fs) per GPU"


PINNED_CPUS = None  # set by main(): the CPUs this rank was given (pin_rank_to_cpus)
JOB_CPUS = None  # set by main() BEFORE pinning: the scheduler affinity of the whole job (restored around the CPU baseline)
JOB_CPU_QUOTA = None  # ... and its CPU quota: what host_cpu_load, summed over the ranks, is to be held against


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--mode", choices=["md", "hrex", "potentials"], default="md")
    ap.add_argument("--frames", type=int, default=4, help="potentials: coordinate frames per execute_batch call")
    ap.add_argument("--systems", default="dhfr,config5", help="potentials: comma-separated subset of dhfr,config5")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--cutoff", type=float, default=1.2)
    ap.add_argument("--workload", choices=["dhfr", "water"], default="dhfr", help="dhfr: 7023 waters + a 2490-atom solute with every bonded term kind and "
                    "scaled 1-4 exclusions (the shape of the reference's DHFR benchmark); water: 7853 waters only (rounds 1-2)")
    ap.add_argument("--no-rc10", action="store_true", help="skip the cutoff-1.0 legs (the reference's own setup_dhfr protocol: rc 1.0, f32)")
    ap.add_argument("--padding", type=float, default=0.18, help="nblist_padding of the nonbonded potential: a speed knob of the potential's constructor, "
                    "results do not depend on it bit for bit (reference default 0.1; 0.18 measured fastest here, DESIGN.md section 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-npt", action="store_true", help="skip the barostat-interval-25 leg (profiling runs: keeps the trace's tail the timed NVT steps)")
    ap.add_argument("--no-rbfe-shape", action="store_true", help="skip the legs on the reference's RBFE state composition (rbfe_shape in the record)")
    ap.add_argument("--profile-steps", type=int, default=400)
    ap.add_argument("--windows", type=int, default=None, help="lambda windows (md: rows of the end-of-run u_kl gather, default 8; hrex: states, default 24)")
    ap.add_argument("--steps-per-frame", type=int, default=400, help="hrex: MD steps between exchanges (fe/free_energy.py default)")
    ap.add_argument("--max-delta-states", type=int, default=4, help="hrex: fe/free_energy.py:99")
    ap.add_argument("--equil-scale", type=float, default=1.0, help="scale the untimed equilibration (profiling runs)")
    ap.add_argument("--equil-precision", choices=["f64", "f32"], default="f32")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto")
    ap.add_argument("--barostat-interval", type=int, default=0, help="hrex: a MonteCarloBarostat every this many steps in every window's context (0: NVT); "
                    "the default f64 NVT line carries an f32 / interval-25 leg (production_shape) unless --no-npt")
    ap.add_argument("--replica-group", type=int, default=4,
                    help="replicas of one rank stepped together on its GPU (custom_ops.multiple_steps_group): hrex mode's MD phase, and the "
                         "replicas_per_gpu legs of md mode; 1 = one after the other, as the reference does")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank drives device 0 (collectives over gloo): a ONE-GPU rehearsal of the N-rank launch -- rank start-up, CPU pinning under the "
                         "job's quota, N HIP contexts, the gather / exchange collectives and grouped stepping on real hardware; the aggregate is one GPU's, not N GPUs'")
    ap.add_argument("--stub", action="store_true", help="no GPU work: a stand-in Context that sleeps (CPU tests of the launch / collective / report plumbing)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# stdout carries exactly ONE line, the JSON record: libraries print banners there (RCCL: "Librccl path : ..." -- it came
# out AFTER the record in a one-rank test), so file descriptor 1 is pointed at stderr for the whole run and the record is
# written to the saved descriptor at the very end
# ---------------------------------------------------------------------------------------------------------------------
_REAL_STDOUT = None


def protect_stdout():
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit_json(record):
    line = json.dumps(record) + "\n"
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line.encode())


# ---------------------------------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------------------------------
def maybe_self_launch(args):
    """--gpus N with no launcher around us: become N ranks (the reference's "one process per device",
    timemachine/parallel/client.py:188-218).  Returns only in the ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.exit(subprocess.call(cmd, env=env))


def init_distributed(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    backend = None
    if world > 1 or os.environ.get("TM_AMD_FORCE_COLLECTIVES"):  # (the variable: one-rank test of the RCCL path on a 1-GPU box)
        import torch
        import torch.distributed as dist

        backend = args.backend
        if backend == "auto":
            backend = "nccl" if (torch.cuda.is_available() and not args.stub and not args.share_gpu) else "gloo"
        if args.share_gpu and backend == "nccl":
            raise SystemExit("bench.py: --share-gpu puts every rank on device 0, which RCCL refuses; use --backend gloo (the default with --share-gpu)")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    return rank, local_rank, world, backend


def device_sync(co):
    if co is not None:
        co.device_synchronize()
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except ImportError:
        pass


WORKLOAD_TEXT = {
    "dhfr": "DHFR-shaped: 7023 flexible TIP3P-like waters + a 2490-atom solute (30 C27H56 chains, amber99-like alkane parameters) = 23559 atoms; "
            "{bonds} bonds, {angles} angles, {torsions} PeriodicTorsion terms (proper n=3/2/1 + improper-form), {exclusions} exclusions of which "
            "{exclusions_14} are 1-4 pairs scaled by (1/1.2, 1/2)",
    "water": "7853 flexible TIP3P-like waters = 23559 atoms; {bonds} bonds, {angles} angles, {torsions} torsions, {exclusions} full exclusions",
}


def term_counts(system):
    partial = int(np.sum(np.any(system.scale_factors != 1.0, axis=1)))
    return {"bonds": len(system.bond_idxs), "angles": len(system.angle_idxs), "torsions": len(system.torsion_idxs),
            "exclusions": len(system.exclusion_idxs), "exclusions_14": partial}


def pci_bus_id(local_rank, stub):
    """PCI bus id of this rank's GPU (None without one): two ranks reporting the same id would share a device"""
    if stub:
        return None
    try:
        import torch

        props = torch.cuda.get_device_properties(local_rank)
        return f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{getattr(props, 'pci_device_id', 0):02x}"
    except Exception:  # pragma: no cover
        return None


def collective_ranks(backend):
    """ranks of the process group the collectives ran in when it is an RCCL ("nccl") group, else None"""
    if backend != "nccl":
        return None
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_initialized() else None


class StubContext:
    """Stand-in for custom_ops.Context (--stub): sleeps 20 us per step.  Exercises everything around the hot path."""

    def __init__(self, n_atoms):
        self.n_atoms = n_atoms
        self._ms = 0.0

    def multiple_steps(self, n, interval=0):
        t0 = time.perf_counter()
        time.sleep(20e-6 * n)
        self._ms = 1e3 * (time.perf_counter() - t0)
        return np.zeros((1, self.n_atoms, 3)), np.zeros((1, 3, 3))

    def last_multiple_steps_ms(self):
        return self._ms

    def get_x_t(self):
        return np.zeros((self.n_atoms, 3))


# ---------------------------------------------------------------------------------------------------------------------
# workload pieces
# ---------------------------------------------------------------------------------------------------------------------
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


def find_all_pairs_of(impl):
    """the NonbondedAllPairs inside one (unbound) potential implementation"""
    for p in _walk(impl):
        if type(p).__name__.startswith("NonbondedAllPairs"):
            return p
    raise RuntimeError("no NonbondedAllPairs in the potential")


def find_nonbonded(bps):
    """the FanoutSummedPotential([AllPairs, Exclusions]) that Nonbonded.to_gpu builds"""
    for bp in bps:
        for p in _walk(bp.get_potential()):
            if type(p).__name__ == "FanoutSummedPotential":
                return p
    raise RuntimeError("no Nonbonded in the state")


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle timed on the host cores; kind "port": a restatement of the reference's JAX path, not JAX)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_quota():
    """CPUs this process may actually use at once: the scheduler affinity, cut down to the cgroup's CPU quota (cpu.max / the v1
    cfs files) where there is one.  The GPU boxes show 256 logical CPUs and grant 16: 64 threads on a 16-CPU quota are throttled
    by the scheduler in bursts -- the cpu_baseline passes of one run then read 44 / 43 / 37 s."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2: "<quota|max> <period>"
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return n


def pin_rank_to_cpus(local_rank, local_world):
    """One process per GPU shares the host: give each rank its own slice of the CPUs this job may use, so that eight ranks' launch
    threads (three launches per ~70 us step each) and their HIP runtime helper threads do not migrate over each other.  The slice
    is cut from the scheduler affinity (the cgroup's cpu.max quota limits CPU TIME, not which CPUs: it is reported in the bench
    line as cpu_quota).  Returns the CPUs of this rank."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return None
    if local_world <= 1 or len(cpus) < 2 * local_world:
        return cpus
    per = len(cpus) // local_world
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:  # pragma: no cover
        return cpus
    return mine


def _cpu_threads(limit=64):
    import torch

    cores = min(_cpu_quota(), limit)  # torch's elementwise kernels stop scaling long before 256 threads
    torch.set_num_threads(cores)
    return cores


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _keep_host_heap():
    """Every torch op of the CPU baseline allocates tens of MB of temporaries; glibc serves requests beyond its mmap threshold
    with mmap / munmap -- a fresh, zero-filled mapping and its page faults per op, serialised on the process's mm lock with
    every worker thread, a third of the CPU time of a pass and the noisiest part of it.  Raise the threshold to its maximum
    (32 MB: the slabs are sized to stay below it) and keep freed memory in the heap: after the untimed passes the same
    pages are reused."""
    import ctypes

    try:
        libc = ctypes.CDLL("libc.so.6")
        ok = libc.mallopt(-3, 1 << 25) and libc.mallopt(-1, (1 << 31) - 1) and libc.mallopt(-2, 1 << 28)  # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD, M_TOP_PAD
        return bool(ok)
    except OSError:
        return False


def cpu_baseline(system, x, cutoff, reps=5, slabs_per_rep=None, rows_per_slab=64):
    """config 3: du/dx of NonbondedAllPairs over the i<j pair matrix, row-blocked dense evaluation (== the reference's JAX
    formulation) in slabs of `rows_per_slab` rows -- by default EVERY slab, i.e. one whole force evaluation per pass, nothing
    extrapolated (`slabs_per_rep` = k: k slabs spread evenly over the triangle, scaled by pair count).  The pass is timed `reps`
    times after two untimed ones (thread pool, heap, clocks); the median is reported, the spread is timing noise.
    Slabs are sized so that (nearly) every temporary stays below glibc's largest mmap threshold (_keep_host_heap): with 384-row
    slabs and the default threshold the same arithmetic read 26-32 s per evaluation instead of 3 -- more kernel time (mmap, page
    faults, munmap per op) than user time, and a spread of 0.07-0.45.  Measured on a GPU box's host (16 threads, whole matrix,
    scripts/cpu_baseline_only.py): 128-row slabs 2.6-2.8 s per evaluation, spread 0.06 (the [128, N, 3] difference tensor is
    72 MB and still mapped per slab: 5-16 s of kernel time per run); 64-row slabs 2.9 s, spread 0.018 (1.5 s)."""
    import torch

    from oracle import ref_potentials as rp

    cores = _cpu_threads()
    heap_kept = _keep_host_heap()
    N = system.num_atoms
    pt = torch.tensor(system.nb_params)
    bt = torch.tensor(system.box)
    pairs_full = N * (N - 1) // 2
    if slabs_per_rep is None:
        starts = list(range(0, N - 1, rows_per_slab))
    else:
        spacing = (N - 1) // slabs_per_rep
        starts = [k * spacing + spacing // 2 for k in range(slabs_per_rep)]
    estimates, seconds = [], []
    for rep in range(-2, reps):  # the SAME slabs every pass: the spread is timing noise only
        xt = torch.tensor(x, requires_grad=True)
        el, pairs_sample = 0.0, 0
        for start in starts:
            r0, r1 = start, min(start + rows_per_slab, N - 1)
            t0 = time.time()
            d3 = rp.delta_r(xt[r0:r1][:, None, :], xt[r0:][None, :, :], torch.diagonal(bt))
            d2 = (d3 * d3).sum(-1)
            upper = torch.arange(r0, r1)[:, None] < torch.arange(r0, N)[None, :]
            d2 = torch.where(upper, d2, torch.full_like(d2, 1e6))
            lj, es = rp._pair_energies(torch.sqrt(d2), pt[r0:r1, 0][:, None] * pt[None, r0:, 0], pt[r0:r1, 1][:, None] + pt[None, r0:, 1], pt[r0:r1, 2][:, None] * pt[None, r0:, 2], system.beta, cutoff)
            (lj.sum() + es.sum()).backward()
            el += time.time() - t0
            pairs_sample += sum(N - 1 - i for i in range(r0, r1))
        if rep < 0:
            continue
        estimates.append(el * pairs_full / pairs_sample)
        seconds.append(el)
    t_step = float(np.median(estimates))
    whole = pairs_sample == pairs_full
    return {
        "value": 86400.0 * DT * 1e-3 / t_step,
        "unit": "ns/day",
        "cores": cores,
        "cpu": _cpu_model(),
        "kind": "port",
        "sample": (f"config 3: du/dx of NonbondedAllPairs over the WHOLE i<j pair matrix in {len(starts)} slabs of {rows_per_slab} rows (one force evaluation per pass, nothing extrapolated)"
                   if whole else f"config 3: du/dx of NonbondedAllPairs on {len(starts)} slabs of {rows_per_slab} rows spread evenly over the i<j pair matrix, scaled to the full matrix by pair count")
        + f", {reps} timed passes after two untimed ones ({sum(seconds):.1f} s of CPU work, median taken); "
        f"torch f64 on {cores} threads{', temporaries served from a kept heap (no mmap per op)' if heap_kept else ''}; oracle restatement of the reference's dense JAX path, not JAX itself",
        "estimates_s": [float(e) for e in estimates],
        "seconds_per_force_eval": t_step,
        "seconds_per_force_eval_extrapolated": t_step,
        "repetitions": reps,
        "spread_rel": float((max(estimates) - min(estimates)) / t_step),
    }


def cpu_baseline_configs():
    """BASELINE configs 1 and 2 on the host cores."""
    from oracle import integrator as oi
    from oracle import ref_potentials as rp
    from timemachine_amd import testsystems as ts

    out = {"cpu": _cpu_model(), "kind": "port"}
    # config 1: 256-atom water cluster, full MD loop (dense forces + python BAOAB), both boxes.  256 x 256 tensors: one
    # thread is the fastest setting by an order of magnitude (64 threads spend their time waking each other up)
    cores1 = _cpu_threads(1)
    for tag, L, n_steps in (("vacuum_100nm", 100.0, 500), ("pbc_3nm", 3.0, 500)):
        s = ts.config1_water_cluster(L)
        N = s.num_atoms
        rng = np.random.default_rng(1)
        ca, cb, cc = oi.langevin_coefficients(TEMPERATURE, 1.0e-3, FRICTION, s.masses)
        x, v = s.coords.copy(), np.zeros((N, 3))
        t0 = time.time()
        for _ in range(n_steps):
            f = -rp.nonbonded(x, s.nb_params, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)[1]
            f -= rp.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs)[1]
            f -= rp.harmonic_angle(x, s.angle_params, s.box, s.angle_idxs)[1]
            x, v = oi.baoab_step(x, v, f, rng.normal(size=(N, 3)), ca, cb, cc, 1.0e-3)
        el = time.time() - t0
        assert np.all(np.isfinite(x))
        out[f"config1_{tag}"] = {"cores": cores1, "steps": n_steps, "seconds": el, "ms_per_step": 1e3 * el / n_steps, "ns_day_at_1fs": n_steps / el * 86400 * 1.0e-3 * 1e-3,
                                 "what": "256 atoms, oracle forces (nonbonded + bond + angle, u and du_dx by autograd) + f64 BAOAB, dt 1 fs"}
    # config 2: ~2.2k-atom solvated ligand, u + du_dx + du_dp of every term
    cores2 = _cpu_threads()
    s = ts.small_solvated_ligand(lamb=0.3)
    x = s.coords
    times = []
    for _ in range(10):
        t0 = time.time()
        rp.nonbonded(x, s.nb_params, s.box, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff)
        rp.harmonic_bond(x, s.bond_params, s.box, s.bond_idxs)
        rp.harmonic_angle(x, s.angle_params, s.box, s.angle_idxs)
        rp.periodic_torsion(x, s.torsion_params, s.box, s.torsion_idxs)
        times.append(time.time() - t0)
    out["config2"] = {"cores": cores2, "repetitions": 10, "seconds_mean": float(np.mean(times)), "seconds_min": float(np.min(times)), "seconds_max": float(np.max(times)),
                      "atoms": s.num_atoms, "what": "u + du_dx + du_dp of Nonbonded, HarmonicBond, HarmonicAngle, PeriodicTorsion (oracle, torch f64 autograd)"}
    return out


def gpu_configs_1_2():
    """The same two workloads as cpu_baseline_configs() on the GPU (f64 potentials), so that every CPU figure has its
    device counterpart in the line: config 1 = ms per MD step of the 256-atom cluster in either box (launch-latency
    bound: three kernels per step), config 2 = seconds per u + du_dx + du_dp evaluation of all four terms (host API:
    includes the host <-> device copies of every call, as the reference's execute() does)."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    out = {"precision": "f64"}
    for tag, L in (("vacuum_100nm", 100.0), ("pbc_3nm", 3.0)):
        s = ts.config1_water_cluster(L)
        bps = [bp.to_gpu(np.float64).bound_impl for bp in ts.bound_potentials(s, np.float64)]
        ctxt = co.Context(s.coords, np.zeros_like(s.coords), s.box, LangevinIntegrator(TEMPERATURE, 1.0e-3, FRICTION, s.masses, 7).impl(), bps)
        ctxt.multiple_steps(500, 0)
        ctxt.multiple_steps(5000, 0)
        out[f"config1_{tag}_ms_per_step"] = ctxt.last_multiple_steps_ms() / 5000
    s = ts.small_solvated_ligand(lamb=0.3)
    pots = [
        (P.Nonbonded(s.num_atoms, s.exclusion_idxs, s.scale_factors, s.beta, s.cutoff).to_gpu(np.float64).unbound_impl, s.nb_params),
        (P.HarmonicBond(s.bond_idxs).to_gpu(np.float64).unbound_impl, s.bond_params),
        (P.HarmonicAngle(s.angle_idxs).to_gpu(np.float64).unbound_impl, s.angle_params),
        (P.PeriodicTorsion(s.torsion_idxs).to_gpu(np.float64).unbound_impl, s.torsion_params),
    ]
    times = []
    for rep in range(12):
        co.device_synchronize()
        t0 = time.time()
        for pot, prm in pots:
            pot.execute(s.coords, prm, s.box, True, True, True)
        co.device_synchronize()
        times.append(time.time() - t0)
    out["config2_seconds_mean"] = float(np.mean(times[2:]))
    out["config2_seconds_min"] = float(np.min(times[2:]))
    return out


# k_nonbonded_tiles<Real, false, true, false>, per dispatch: (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 -- the gfx950
# correction of MI355X_MICROARCH.md's HBM section doubles FETCH_SIZE.  Fallback when profiles/pmc_traffic.json is absent.
PMC_TRAFFIC = {
    "f64": {"bytes": (2 * 11259 + 46582) * 1024, "source": "profiles/r02_v1_pmc_md_f64.txt"},
    "f32": {"bytes": (2 * 7413 + 34323) * 1024, "source": "profiles/r02_v1_pmc_md_f32.txt"},
}


def load_pmc_traffic():
    """newest committed PMC summary wins (profiles/pmc_traffic.json is written by scripts/gpu_pmc.sh's post-processing)"""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return PMC_TRAFFIC


# ---------------------------------------------------------------------------------------------------------------------
# --mode md
# ---------------------------------------------------------------------------------------------------------------------
def run_md(args, rank, local_rank, world, backend):
    from timemachine_amd import parallel

    co = None
    if not args.stub:
        from timemachine_amd import potentials as P
        from timemachine_amd import testsystems as ts
        from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

        if co.device_count() < 1:
            raise SystemExit("bench.py needs a GPU: timemachine_amd has no CPU fallback")
        co.set_device(0 if args.share_gpu else local_rank)
        precision = np.float64 if args.precision == "f64" else np.float32
        build = ts.dhfr_shaped_box if args.workload == "dhfr" else ts.dhfr_sized_water_box
        system = build(seed=2025, hmr=True, cutoff=args.cutoff)
        N = system.num_atoms

        def make_bps(prec, cutoff=None):
            # one SummedPotential for the whole state -- how the reference packs a state
            # (fe/free_energy.py:614-657: make_summed_potential(...).to_gpu(np.float32) -> one BoundPotential)
            sys_c = system if cutoff is None else dataclasses.replace(system, cutoff=cutoff)
            bps = ts.bound_potentials(sys_c, prec, nblist_padding=args.padding)
            summed = P.SummedPotential([bp.potential for bp in bps], [bp.params for bp in bps])
            return [summed.bind_params_list([bp.params for bp in bps]).to_gpu(prec).bound_impl]

        seed = 1234 + rank
        x, v = equilibrate(co, LangevinIntegrator, system, make_bps, seed, args.equil_scale, np.float64 if args.equil_precision == "f64" else np.float32)
    else:
        N, system = 23559, None

    rank_ms_per_step = []  # one entry per run() call; [0] is the headline run
    rank_cpu_us_per_step = []  # process CPU time of the timed multiple_steps call / steps, same indexing

    def run(prec, steps, warmup, profile_steps, barostat_interval=0, cutoff=None, start=None, final=None):
        # start: (x, v, box) to begin from instead of the equilibrated frame; final: a dict that receives the run's last (x, v, box)
        x_s, v_s, box_s = start if start is not None else ((None, None, None) if args.stub else (x, v, system.box))
        if args.stub:
            bps, ctxt = None, StubContext(N)
        else:
            bps = make_bps(prec, cutoff)
            movers = []
            if barostat_interval > 0:
                # the reference's second DHFR line: "dhfr-apo-barostat-interval-25" (tests/test_benchmark.py:517-518, 222-232)
                from timemachine_amd.lib import MonteCarloBarostat

                movers = [MonteCarloBarostat(N, 1.0, TEMPERATURE, ts.molecule_groups(system), barostat_interval, seed).impl(bps)]
            ctxt = co.Context(x_s, v_s, box_s, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, seed).impl(), bps, movers=movers)
        device_sync(co)  # the first call initialises torch's HIP context (seconds): do it here, not in front of the clock
        ctxt.multiple_steps(SETTLE_STEPS, 0)  # untimed, whatever --warmup says (see the module docstring)
        if warmup > 0:
            ctxt.multiple_steps(warmup, 0)
        parallel.barrier()
        device_sync(co)
        t0 = time.perf_counter()
        c0 = time.process_time()  # user + system CPU time of this process, all threads (the HIP runtime's included)
        ctxt.multiple_steps(steps, 0)
        dev_s = 1e-3 * ctxt.last_multiple_steps_ms()
        device_sync(co)
        rank_cpu_us_per_step.append(1e6 * (time.process_time() - c0) / steps)
        parallel.barrier()
        host_s = time.perf_counter() - t0
        rank_ms_per_step.append(1e3 * dev_s / steps)  # this rank's own clock, before the max over ranks
        dev_s, host_s = parallel.max_over_ranks(dev_s), parallel.max_over_ranks(host_s)
        xf = ctxt.get_x_t()
        assert np.all(np.isfinite(xf)), "trajectory diverged"
        if final is not None and not args.stub:
            final.update(x=xf, v=ctxt.get_v_t(), box=ctxt.get_box())
        prof = None
        if profile_steps > 0 and not args.stub:
            nb = find_all_pairs(bps)
            builds0 = nb.get_build_count()
            co.profile_reset()
            co.profile_set_enabled(True)
            ctxt.multiple_steps(profile_steps, 0)
            total_ms, launches = co.profile_read("nonbonded_tiles")
            per_kernel = {name: co.profile_read(name) for name in ("nonbonded_tiles", "nblist_build", "integrator_update")}
            co.profile_set_enabled(False)
            co.profile_reset()
            tiles = nb.get_tile_ixn_count()
            for _ in range(8):  # the list counters read 0 between the step that asked for a rebuild and the rebuild itself
                if tiles:
                    break
                ctxt.multiple_steps(1, 0)
                tiles = nb.get_tile_ixn_count()
            prof = {"kernel_ms": total_ms / max(launches, 1), "launches": launches, "tiles": tiles,
                    "run_ms": ctxt.last_multiple_steps_ms(), "steps": profile_steps, "builds": nb.get_build_count() - builds0,
                    "per_kernel": per_kernel, "listed_atoms": nb.get_tile_ixn_count() * 32}
        return dev_s, host_s, xf, bps, prof

    dev_s, host_s, xf, bps, prof = run(None if args.stub else precision, args.steps, args.warmup, args.profile_steps if rank == 0 else 0)
    ns_day = args.steps / dev_s * 86400.0 * DT * 1e-3 * world

    # ---- end-of-run reduced-potential gather (outside the timed region): each rank evaluates its final frame under
    # every window's parameters (charges scaled by lambda_l) and the rows are all-gathered over RCCL
    n_windows = max(args.windows or 8, world)
    my_windows = parallel.windows_for_rank(world, world, rank)  # one replica per rank
    if args.stub:
        u_row = np.full((1, n_windows), float(rank))
    else:
        lambdas = np.linspace(0.0, 1.0, n_windows)
        params_l = np.stack([system.nb_params * np.array([1.0 - 0.1 * lam, 1.0, 1.0, 1.0]) for lam in lambdas])
        _, _, u_row = find_nonbonded(bps).execute_batch(xf[None], params_l, system.box[None], False, False, True)
        u_row = u_row.reshape(1, -1) / (0.008314462618 * TEMPERATURE)
    parallel.barrier()
    t0 = time.perf_counter()
    u_kl = parallel.gather_rows(my_windows, u_row, world, row_length=n_windows)
    gather_ms = 1e3 * parallel.max_over_ranks(time.perf_counter() - t0)
    gather_ok = bool(np.all(np.isfinite(u_kl))) and u_kl.shape == (world, n_windows)
    # who ran what, so that an N > 1 record explains itself: every rank's own ms per step, device and bus id
    per_rank = parallel.gather_objects({
        "rank": rank, "local_rank": local_rank, "ms_per_step": rank_ms_per_step[0], "windows": my_windows,
        "host_cpu_us_per_step": rank_cpu_us_per_step[0], "cpus": len(PINNED_CPUS) if PINNED_CPUS else None,
        "device": "stub" if args.stub else co.device_name(), "pci_bus_id": pci_bus_id(0 if args.share_gpu else local_rank, args.stub), "host": socket.gethostname(),
    })

    if rank != 0:
        return

    out = {
        "metric": METRIC,
        "value": ns_day,
        "unit": "ns/day",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dev_s / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": "configs[2] (zero-based; BASELINE's third): DHFR-sized explicit-solvent box, "
            + (WORKLOAD_TEXT[args.workload] if args.stub else WORKLOAD_TEXT[args.workload].format(**term_counts(system))) +
            f", box 6.223 nm, direct-space erfc*switch electrostatics + LJ, cutoff {args.cutoff} nm, beta 2.0, HMR, dt 2.5 fs, Langevin 300 K friction 1/ps; "
            "one independent replica per GPU",
            "atoms": N,
            "terms": None if args.stub else term_counts(system),
            "potentials": ("HarmonicBond/HarmonicAngle/" + ("PeriodicTorsion/" if args.stub or len(system.torsion_idxs) else "")
                           + f"Nonbonded(AllPairs+Exclusions) *_{args.precision} packed into one SummedPotential, LangevinIntegrator<float>"),
            "replicas": world,
            "nblist_padding": args.padding,
        },
        "timing": f"HIP events on the Context stream around the {args.steps} timed steps (max over ranks), after {SETTLE_STEPS} settle + {args.warmup} warm-up steps",
        "host_ms_per_step": 1e3 * host_s / args.steps,
        # what the host spends to drive one step (process CPU time of the timed multiple_steps call / steps: the launch thread and the
        # HIP runtime's helpers), the CPUs the job may use (scheduler affinity cut to the cgroup quota) and what N ranks of this kind
        # ask of them: host_cpu_load = n_gpus * host_cpu_us_per_step / (1e3 * ms_per_step) CPUs busy, to be held against cpu_quota
        "host_cpu_us_per_step": max(r["host_cpu_us_per_step"] for r in per_rank),
        "cpu_quota": JOB_CPU_QUOTA if JOB_CPU_QUOTA is not None else _cpu_quota(),  # the whole job's (taken before the ranks pinned themselves)
        "host_cpu_load": sum(r["host_cpu_us_per_step"] for r in per_rank) / (1e3 * (1e3 * dev_s / args.steps)),
        "host_ns_day": args.steps / host_s * 86400.0 * DT * 1e-3 * world,
        "world_size": world,
        "backend": backend,
        # `value` is the AGGREGATE over the n_gpus replicas (the contract's whole-job throughput); the metric's "per GPU" figure:
        "value_per_gpu": ns_day / world,
        "rccl_ranks": collective_ranks(backend),
        "per_rank": per_rank,
        "binding": "stub" if args.stub else co.BINDING,
        "device": "stub" if args.stub else co.device_name(),
        "mbar_gather_ok": gather_ok,
        "mbar_gather_ms": gather_ms,
    }
    if args.share_gpu:
        out["share_gpu"] = True
        out["share_gpu_note"] = (f"REHEARSAL: all {world} ranks drive device 0 (gloo collectives); `value` is ONE GPU's aggregate over {world} concurrently stepped "
                                 "processes -- compare it with replicas_per_gpu of the one-process line, not with an N-GPU figure")

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
        pmc = load_pmc_traffic().get(args.precision, {})
        out["profiled"] = {"steps": prof["steps"], "ns_day": prof["steps"] / (1e-3 * prof["run_ms"]) * 86400.0 * DT * 1e-3,
                           "steps_per_list_build": prof["steps"] / max(prof["builds"], 1), "note": "the same Context, per-launch HIP events on"}
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "k_nonbonded_tiles",
            "achieved": bytes_alg / t_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": bytes_alg / t_s / 1e9 / HBM_PEAK_GBS,
            # 2 * FETCH_SIZE + WRITE_SIZE per dispatch of this kernel from the committed PMC passes (separate rocprofv3 --pmc
            # runs of this very command, scripts/gpu_pmc.sh).  Most of it is the flush: device-scope u64 atomics are executed
            # at the memory side of the fabric (the eight XCD L2s are not coherent), so every one is a write request.
            "traffic": pmc.get("bytes"),
            "traffic_source": f"{pmc.get('source')} (2 * FETCH_SIZE + WRITE_SIZE, KB per dispatch, gfx950 correction applied; measured in separate --pmc passes, not in this run)",
            "bytes_per_launch": bytes_alg,
            "kernel_ms": prof["kernel_ms"],
            "launches_timed": prof["launches"],
            "note": "required HBM form; this kernel is VALU-bound, not HBM-bound (SURVEY.md F10) -- see roofline_valu",
        }
        # the whole step, kernel by kernel (per-launch HIP events of the profiled steps: each bracket includes ~1-3 us of dispatch,
        # so the small kernels read larger here than in rocprofv3's kernel trace, profiles/r04_*_per_step_*.txt)
        steps_p = max(prof["steps"], 1)
        real = 8 if args.precision == "f64" else 4
        builds = max(prof["builds"], 1)
        alg_bytes = {
            # per launch: one gathered record read + one u64 x 3 accumulator RMW per atom; 4 B id + 128 B column indices per tile
            "nonbonded_tiles": bytes_alg,
            # per REBUILD (amortised below): x y z of every atom once per row block that lists it is L2 traffic; from HBM: the
            # records once (3 * sizeof(Real) * N), block bounds (2 x 3 x sizeof(Real) per 32 atoms), 4 B per listed column atom
            # written, 16 B per 64-column work item, the snapshot copy (48 B per atom)
            "nblist_build": (3 * real + 48) * N + 6 * real * (N // 32 + 1) + 4 * prof["listed_atoms"] + 16 * (prof["listed_atoms"] // 64 + 1),
            # per launch (sorted hand-over form): read perm 4, x v (slot order) 48, cb cc 8, sorted accumulator 24, snapshot 24;
            # write x v twice (atom order + slot order) 96, the sorted record's x y z 3 * sizeof(Real), the accumulator's zeros 24
            "integrator_update": (4 + 48 + 8 + 24 + 24 + 96 + 3 * real + 24) * N,
        }
        out["kernels"] = []
        for name, (ms_k, n_k) in prof["per_kernel"].items():
            if not n_k:
                continue
            us_per_step = 1e3 * ms_k / steps_p
            b = alg_bytes[name] * (builds if name == "nblist_build" else n_k) / steps_p  # algorithmic bytes per STEP
            out["kernels"].append({
                "name": name, "us_per_step": us_per_step, "launches_per_step": n_k / steps_p, "us_per_launch": 1e3 * ms_k / n_k,
                "algorithmic_bytes_per_step": b, "frac_of_hbm_peak": b / (us_per_step * 1e-6) / 1e9 / HBM_PEAK_GBS,
                # of the PROFILED step -- the same run and clock the brackets come from (the profiled steps are slower than the timed
                # ones by the brackets' own dispatch latency; dividing by the timed step made the shares sum to 1.12)
                "share_of_step": us_per_step / (1e3 * prof["run_ms"] / steps_p),
            })
        out["kernels_profiled_us_per_step"] = 1e3 * prof["run_ms"] / steps_p
        out["kernels_note"] = ("per-launch HIP events on the Context stream over the profiled steps; nblist_build = every launch of the list kernels "
                               "(rebuilds, amortised, and the launches that only read the rebuild flag); frac_of_hbm_peak uses the algorithmic bytes "
                               "spelt out in bench.py (alg_bytes) -- none of these kernels is HBM-bound: the tile kernel is bound by VALU issue "
                               "(roofline_valu), the other two by launch and dependent-load latency")
        # the committed PMC summary belongs to ONE build of the library: say so when this run's library is another
        stamp = None
        try:
            with open(os.path.join(REPO, "timemachine_amd", "csrc", ".build_stamp")) as fh:
                stamp = fh.read().strip()
        except OSError:
            pass
        out["roofline"]["traffic_build_stamp"] = pmc.get("build_stamp")
        out["roofline"]["library_build_stamp"] = stamp
        # (a PMC record without a stamp cannot vouch for any build: stale)
        out["roofline"]["traffic_stale"] = (pmc.get("build_stamp") != stamp) if stamp else None
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
            "kernel_share_of_step": prof["kernel_ms"] / (1e3 * dev_s / args.steps),
            "flop_constants": {"per_pair": C_PAIR_FLOPS, "per_slot": C_SLOT_FLOPS, "source": "SURVEY.md section 8(d): the reference formula's operation count"},
        }
        # the honest instruction picture: this kernel's own flop count, and what the SQ counters of the committed PMC passes say
        own = OWN_PAIR_FLOPS[args.precision] * p_int + OWN_SLOT_FLOPS * 1024 * tiles
        rv = out["roofline_valu"]
        rv["own_isa"] = {"flops_per_pair": OWN_PAIR_FLOPS[args.precision], "flops_per_slot": OWN_SLOT_FLOPS, "flops_per_launch": own,
                         "achieved": own / t_s / 1e12, "frac": own / t_s / 1e12 / peak_fl,
                         "note": "flops this kernel's ISA executes per pair / per filter slot (FMA = 2); the fraction of peak a flop count can reach is "
                                 "bounded by the share of FMA among the issued instructions -- the kernel is bound by instruction ISSUE (all types), see insts_all_per_launch / valu_busy"}
        sq = pmc.get("sq") or {}
        if sq.get("SQ_INSTS_VALU"):
            kernel_cycles = t_s * CLOCK_GHZ * 1e9
            rv["insts_valu_per_launch"] = sq["SQ_INSTS_VALU"]
            rv["lane_ops_per_pair"] = sq["SQ_INSTS_VALU"] * 64.0 / p_int
            # SQ_ACTIVE_INST_VALU counts quad-cycles the VALUs spend executing; 1024 SIMDs x kernel cycles are available
            rv["valu_busy"] = sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * kernel_cycles)
            rv["insts_all_per_launch"] = sq.get("SQ_INSTS")  # every instruction type: the kernel is bound by issue, not by VALU time alone
            rv["insts_salu_per_launch"] = sq.get("SQ_INSTS_SALU")
            rv["insts_lds_per_launch"] = sq.get("SQ_INSTS_LDS")
            rv["lds_bank_conflict_cycles"] = sq.get("SQ_LDS_BANK_CONFLICT")
            if sq.get("SQ_ACTIVE_INST_LDS"):
                # the CU's ONE LDS pipe: SQ_ACTIVE_INST_LDS in the units a saturated pipe reads -- scripts/microbench/lds_atomics (sixteen
                # waves per CU issuing nothing but LDS instructions) reads 1.84-1.98 by the valu_busy formula taken per CU
                # (scripts/gpu_lds_counter_check.sh), i.e. the counter ticks every ~2.1 cycles
                rv["lds_busy"] = sq["SQ_ACTIVE_INST_LDS"] * 4.0 / (256.0 * kernel_cycles) / LDS_COUNTER_SATURATED
                rv["lds_busy_note"] = (f"SQ_ACTIVE_INST_LDS * 4 / (256 CUs * kernel cycles) / {LDS_COUNTER_SATURATED} (what a saturated pipe reads: "
                                       "scripts/gpu_lds_counter_check.sh); bank-conflict cycles of 32-bit atomics are not in the counter, so a lower bound")
            rv["counters_source"] = f"{pmc.get('source')} (separate rocprofv3 --pmc passes of this command; valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel_ms * {CLOCK_GHZ} GHz))"

    if world == 1 and not args.stub:
        # the other precision, for the record (the reference ships f32 kernels; BASELINE asks for f64 forces)
        # (secondary legs are not `value`: they run SECONDARY_STEPS timed steps each -- 20 steps hold 1 to 3 list rebuilds)
        other = np.float32 if precision == np.float64 else np.float64
        n_sec = max(args.steps, SECONDARY_STEPS)
        w_sec = max(args.warmup, 50)
        tag = {np.float32: "f32", np.float64: "f64"}
        try:
            d2, _, _, _, _ = run(other, n_sec, w_sec, 0)
            out["ns_day_" + tag[other]] = n_sec / d2 * 86400.0 * DT * 1e-3
        except Exception as exc:  # pragma: no cover
            out["other_precision_error"] = str(exc)
        # the reference's own protocol line: setup_dhfr uses cutoff 1.0 nm (testsystems/dhfr.py:21) and its benchmark f32
        # potentials (tests/test_benchmark.py:219,504-518); same coordinates, same padding, both precisions
        if not args.no_rc10 and args.cutoff != 1.0:
            try:
                rc10 = {}
                for prec in (np.float32, np.float64):
                    d4, _, _, _, _ = run(prec, n_sec, w_sec, 0, cutoff=1.0)
                    rc10[tag[prec]] = n_sec / d4 * 86400.0 * DT * 1e-3
                out["rc1.0_f32"], out["rc1.0_f64"] = rc10["f32"], rc10["f64"]
                out["rc1.0_note"] = (f"ns/day of the same box at cutoff 1.0 nm, {n_sec} timed steps each; rc1.0_f32 is the configuration the reference's "
                                     "dhfr-apo benchmark runs (testsystems/dhfr.py:21, tests/test_benchmark.py:219)")
            except Exception as exc:  # pragma: no cover
                out["rc10_error"] = str(exc)
        # NPT: the same box with the Monte Carlo barostat every 25 steps (the reference benchmarks both ensembles)
        try:
            if args.no_npt:
                raise KeyboardInterrupt
            n_npt = n_sec
            end = {}
            d3, _, _, _, _ = run(precision, n_npt, w_sec, 0, barostat_interval=25, final=end)
            out["npt"] = {"barostat_interval": 25, "pressure_bar": 1.0, "ns_day": n_npt / d3 * 86400.0 * DT * 1e-3,
                          "ms_per_step": 1e3 * d3 / n_npt, "dtype": args.precision,
                          "note": "reference: tests/test_benchmark.py:517-518 (dhfr-apo-barostat-interval-25); every attempt evaluates both energies "
                                  "in ONE tile launch on the nonbonded potential's current list (csrc/barostat.hip, the fast path)"}
            # At 1 bar this synthetic box contracts (6.223 -> ~6.14 nm: its water model's equilibrium density is not real water's), so
            # the NPT steps do ~4 % more pair work than `value`'s at the DHFR box.  What the BAROSTAT costs is the ratio against an NVT
            # run of the same state -- the NPT run's last frame, box and velocities, the same number of steps:
            d5, _, _, _, _ = run(precision, n_npt, 0, 0, start=(end["x"], end["v"], end["box"]))
            out["npt"]["box_nm_at_end"] = float(end["box"][0, 0])
            out["npt"]["nvt_at_npt_box_ns_day"] = n_npt / d5 * 86400.0 * DT * 1e-3
            out["npt"]["ratio_to_nvt_at_npt_box"] = d5 / d3
            out["npt"]["ratio_to_value"] = (n_npt / d3) / (args.steps / dev_s)
        except KeyboardInterrupt:
            pass
        except Exception as exc:  # pragma: no cover
            out["npt_error"] = str(exc)
        # several independent replicas of the same box on this ONE GPU, stepped together (free-energy windows / HREX replicas that
        # share a device): NOT `value` -- that is one trajectory, as the reference's benchmark times it -- but what the device delivers
        # per day when it has more than one window to run
        if not args.no_npt and args.replica_group > 1:
            try:
                out["replicas_per_gpu"] = {}
                # (the last leg: the reference benchmark's own configuration -- f32 potentials, cutoff 1.0 nm -- at the full group size)
                legs = [(n_rep, precision, None, str(n_rep)) for n_rep in sorted({2, args.replica_group})]
                if not args.no_rc10 and args.cutoff != 1.0:
                    legs.append((args.replica_group, np.float32, 1.0, f"{args.replica_group}_rc1.0_f32"))
                for n_rep, leg_prec, leg_cutoff, key in legs:
                    group = [co.Context(x, v, system.box, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, seed + 17 * k).impl(), make_bps(leg_prec, leg_cutoff))
                             for k in range(n_rep)]
                    co.multiple_steps_group(group, SETTLE_STEPS)
                    device_sync(co)
                    t0 = time.perf_counter()
                    c0 = time.process_time()
                    co.multiple_steps_group(group, n_sec)
                    wall = time.perf_counter() - t0
                    cpu_s = time.process_time() - c0
                    assert all(np.all(np.isfinite(c.get_x_t())) for c in group), "trajectory diverged"
                    out["replicas_per_gpu"][key] = {
                        "replicas": n_rep, "aggregate_ns_day": n_rep * n_sec / wall * 86400.0 * DT * 1e-3, "us_per_replica_step": 1e6 * wall / (n_sec * n_rep),
                        "device_ms_per_step_each": [c.last_multiple_steps_ms() / n_sec for c in group],
                        "dtype": "f64" if leg_prec == np.float64 else "f32", "cutoff": args.cutoff if leg_cutoff is None else leg_cutoff,
                        # the host side of the grouped call: process CPU time per replica-step, CPUs busy during the call (hold n_gpus
                        # times this against cpu_quota), the enqueueing threads and the hardware queues the runtime was given
                        "host_cpu_us_per_step": 1e6 * cpu_s / (n_sec * n_rep), "host_cpu_load": cpu_s / wall,
                        "enqueue_threads": _group_threads(N), "gpu_max_hw_queues": _runtime_env("GPU_MAX_HW_QUEUES")}
                    del group
                out["replicas_per_gpu"]["note"] = (
                    "independent replicas of the same box stepped together on one GPU (custom_ops.multiple_steps_group: one enqueueing host thread at this size; steps "
                    "interleaved on the contexts' own streams; one replica's list / update kernels run underneath another's force kernel); "
                    "host wall clock of the call; trajectories bit-identical to stepping alone (tests/test_gpu_parity.py)")
            except Exception as exc:  # pragma: no cover
                out["replicas_per_gpu_error"] = str(exc)
        if not args.no_rbfe_shape:
            try:
                out["rbfe_shape"] = rbfe_shape_legs(co, args, seed, n_sec)
            except Exception as exc:  # pragma: no cover
                out["rbfe_shape_error"] = repr(exc)
        if not args.no_cpu_baseline:
            if JOB_CPUS:  # the CPU baseline uses the job's CPUs, not this rank's slice of them
                try:
                    os.sched_setaffinity(0, JOB_CPUS)
                except OSError:  # pragma: no cover
                    pass
            out["cpu_baseline"] = cpu_baseline(system, xf, args.cutoff)
            out["cpu_baseline_configs"] = cpu_baseline_configs()
            if not args.stub:
                out["cpu_baseline_configs"]["gpu_beside"] = gpu_configs_1_2()

    emit_json(out)


# ---------------------------------------------------------------------------------------------------------------------
# the composition the reference's RBFE windows have (fe/system.py:133-146), next to the benchmark's single Nonbonded
# ---------------------------------------------------------------------------------------------------------------------
def rbfe_shape_legs(co, args, seed, n_sec, sizes=("config4", "config5")):
    """What the engine delivers on the state composition production runs (HostGuestSystem: bonded terms, chiral restraints,
    ligand-ligand precomputed pairs, host-host Nonbonded(atom_idxs=host), ligand-environment NonbondedInteractionGroup -- packed into
    one SummedPotential as fe/free_energy.py:614-657 packs it) against the SAME box with one all-atom Nonbonded (the composition of the
    reference's dhfr / hif2a BENCHMARK states and of `value`), at BASELINE config 4's size (6.3k atoms, tests/test_benchmark.py:541)
    and config 5's (31.5k): NVT in both precisions, NPT with the barostat every 25 steps (and which path its attempts took), four
    windows stepped together, and the composition with producer merging switched off (rounds 1-5: two lists, two tile launches, the
    atom-order update kernel)."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat

    out = {}
    for size in sizes:
        system, n_lig = (ts.config4_solvated_ligand(0.3), 30) if size == "config4" else (ts.config5_complex_sized(0.3), 40)
        N = system.num_atoms

        def pack(bound, prec):
            summed = P.SummedPotential([bp.potential for bp in bound], [bp.params for bp in bound])
            return [summed.bind_params_list([bp.params for bp in bound]).to_gpu(prec).bound_impl]

        def make_bps(prec, composition="single"):
            bound = ts.bound_potentials(system, prec, nblist_padding=args.padding) if composition == "single" else ts.rbfe_bound_potentials(system, n_lig, nblist_padding=args.padding)
            return pack(bound, prec)

        x, v = equilibrate(co, LangevinIntegrator, system, make_bps, seed, args.equil_scale, np.float32)

        def leg(prec, composition, barostat_interval=0, merge=True, n_group=1):
            before = co.debug_set_merge_producers(merge)
            try:
                ctxts, baros, all_bps = [], [], []
                for k in range(n_group):
                    bps = make_bps(prec, composition)
                    movers = [MonteCarloBarostat(N, 1.0, TEMPERATURE, ts.molecule_groups(system), barostat_interval, seed + k).impl(bps)] if barostat_interval else []
                    ctxts.append(co.Context(x, v, system.box, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, seed + 31 * k).impl(), bps, movers=movers))
                    baros += movers
                    all_bps.append(bps)
                rec = {}
                if n_group == 1:
                    ctxt = ctxts[0]
                    ctxt.multiple_steps(SETTLE_STEPS, 0)
                    device_sync(co)
                    ctxt.multiple_steps(n_sec, 0)
                    dev_s = 1e-3 * ctxt.last_multiple_steps_ms()
                    rec.update(ns_day=n_sec / dev_s * 86400.0 * DT * 1e-3, us_per_step=1e6 * dev_s / n_sec)
                else:
                    co.multiple_steps_group(ctxts, SETTLE_STEPS)
                    device_sync(co)
                    t0 = time.perf_counter()
                    co.multiple_steps_group(ctxts, n_sec)
                    wall = time.perf_counter() - t0
                    rec.update(aggregate_ns_day=n_group * n_sec / wall * 86400.0 * DT * 1e-3, us_per_replica_step=1e6 * wall / (n_sec * n_group), replicas=n_group)
                assert all(np.all(np.isfinite(c.get_x_t())) for c in ctxts), "trajectory diverged"
                if baros:
                    attempts, fast = baros[0].get_attempt_paths()
                    rec.update(barostat_attempts=attempts, barostat_attempts_on_current_list=fast)
                nb = find_all_pairs(all_bps[0])
                calls, tiles, builds = nb.get_merged_stats()
                rec["merged_evaluations"] = calls  # evaluations the all-pairs potential made as the carrier of the interaction group
                for _ in range(8):  # (the list counters read 0 between the step that asked for a rebuild and the rebuild itself)
                    tiles = nb.get_merged_stats()[1] if calls else nb.get_tile_ixn_count()
                    if tiles:
                        break
                    ctxts[0].multiple_steps(1, 0)
                rec["tiles_32x32"] = tiles
                return rec
            finally:
                co.debug_set_merge_producers(before)

        f64, f32 = np.float64, np.float32
        r = {"atoms": N, "ligand_atoms": n_lig, "timed_steps": n_sec}
        r["single_nonbonded"] = {"nvt_f64": leg(f64, "single"), "nvt_f32": leg(f32, "single"), "npt_25_f32": leg(f32, "single", 25), "grouped_4_f32": leg(f32, "single", n_group=4)}
        r["nvt_f64"] = leg(f64, "rbfe")
        r["nvt_f32"] = leg(f32, "rbfe")
        r["npt_25_f32"] = leg(f32, "rbfe", 25)
        r["npt_25_f64"] = leg(f64, "rbfe", 25)
        r["grouped_4_f32"] = leg(f32, "rbfe", n_group=4)
        r["producers_not_merged"] = {"nvt_f64": leg(f64, "rbfe", merge=False), "nvt_f32": leg(f32, "rbfe", merge=False), "npt_25_f32": leg(f32, "rbfe", 25, merge=False)}
        r["ratio_to_single_nonbonded"] = {k: r[k]["ns_day"] / r["single_nonbonded"][k]["ns_day"] for k in ("nvt_f64", "nvt_f32", "npt_25_f32")}
        r["ratio_to_single_nonbonded"]["grouped_4_f32"] = r["grouped_4_f32"]["aggregate_ns_day"] / r["single_nonbonded"]["grouped_4_f32"]["aggregate_ns_day"]
        r["npt_over_nvt_f32"] = r["npt_25_f32"]["ns_day"] / r["nvt_f32"]["ns_day"]
        r["npt_over_nvt_f64"] = r["npt_25_f64"]["ns_day"] / r["nvt_f64"]["ns_day"]
        out[size] = r
    out["note"] = ("HostGuestSystem.get_U_fns() (timemachine/fe/system.py:133-146; testsystems.rbfe_shaped_state) packed into one SummedPotential, "
                   "against one all-atom Nonbonded on the same box (single_nonbonded); HIP events around the timed steps of one trajectory, host wall "
                   "clock for the grouped legs; merged_evaluations > 0: the two tile producers ran as one pipeline (csrc/engine.hpp, merged carrier); "
                   "producers_not_merged: the same state with tm_debug_set_merge_producers(0) -- rounds 1-5's path")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# --mode potentials: the reference benchmark's second half (tests/test_benchmark.py:148-191, 624-678: benchmark_potential)
# ---------------------------------------------------------------------------------------------------------------------
def run_potentials(args):
    """`benchmark_potential`: unbound.execute_batch(coords[F], params[5], boxes[F], du_dx, du_dp, u) timed host to host, executions per
    second, f32 and f64 -- for Nonbonded, NonbondedInteractionGroup (the two the reference singles out, :624-678) and the SummedPotential
    of the reference's RBFE composition (testsystems.rbfe_shaped_state), on the DHFR-shaped box (23.5k atoms) and the config-5-sized
    complex (31.4k).  Frames: F snapshots 10 MD steps apart (:103).  Parameter sets: five copies of the state's (as the reference
    stacks them, :647,:664) AND five lambda windows (ligand charges / w scaled: what compute_potential_matrix and u_kln
    re-evaluation feed, fe/free_energy.py:1148-1200).  Beside the full call (u + du/dx + du/dp): forces only, energy only, and a single
    execute() of every term of config 2 (2 243 atoms)."""
    from timemachine_amd import potentials as P
    from timemachine_amd import testsystems as ts
    from timemachine_amd.lib import LangevinIntegrator, custom_ops as co

    if co.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: timemachine_amd has no CPU fallback")
    co.set_device(0)
    n_frames, n_params, n_batches = args.frames, 5, 2
    out = {"metric": "potential executions per second, host to host (execute_batch over frames x 5 parameter sets)", "unit": "executions/s", "frames": n_frames,
           "param_sets": n_params, "batches_timed": n_batches, "reference": "tests/test_benchmark.py:148-191 (benchmark_potential), :624-678", "device": co.device_name(),
           "binding": co.BINDING, "systems": {}}

    def frames_of(system, make_bound):
        bps = [bp.to_gpu(np.float32).bound_impl for bp in make_bound()]
        x, v = equilibrate(co, LangevinIntegrator, system, lambda p: [bp.to_gpu(p).bound_impl for bp in make_bound()], 11, args.equil_scale, np.float32)
        ctxt = co.Context(x, v, system.box, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, 3).impl(), bps)
        xs, boxes = ctxt.multiple_steps(10 * n_frames, 10)
        return xs, boxes

    def throttled_us():
        """CPU time this job was denied by its cgroup quota so far (cpu.stat: throttled_usec), or None"""
        try:
            with open("/sys/fs/cgroup/cpu.stat") as fh:
                for line in fh:
                    if line.startswith("throttled_usec"):
                        return int(line.split()[1])
        except (OSError, ValueError):
            pass
        return None

    def timed(unbound, coords, params, boxes, flags):
        for _ in range(int(os.environ.get("TM_AMD_POT_WARMUPS", "1"))):
            unbound.execute_batch(coords, params, boxes, *flags)  # untimed: allocations, list builds, clocks
        runs = coords.shape[0] * params.shape[0]
        ts_ = []
        th0, c0 = throttled_us(), time.process_time()
        dev_ms = []
        for _ in range(n_batches):
            co.device_synchronize()
            t0 = time.perf_counter()
            unbound.execute_batch(coords, params, boxes, *flags)
            ts_.append(time.perf_counter() - t0)
            dev_ms.append(co.debug_last_host_call_device_ms())
        th1 = throttled_us()
        return {"executions_per_s": runs / float(np.mean(ts_)), "us_per_execution": 1e6 * float(np.mean(ts_)) / runs, "us_per_execution_best_batch": 1e6 * float(np.min(ts_)) / runs,
                # the evaluations alone on the device (HIP events behind the staging copy and in front of the conversion / copy back)
                "device_us_per_execution": 1e3 * float(np.mean(dev_ms)) / runs,
                "host_cpu_s_per_wall_s": (time.process_time() - c0) / max(sum(ts_), 1e-9), "cgroup_throttled_us": None if th0 is None or th1 is None else th1 - th0}

    forms = {"u_du_dx_du_dp": (True, True, True), "du_dx": (True, False, False), "u": (False, False, True)}
    for name in args.systems.split(","):
        if name == "dhfr":
            system, n_lig = ts.dhfr_shaped_box(), 0
        else:
            system, n_lig = ts.config5_complex_sized(0.3), 40
        N = system.num_atoms
        xs, boxes = frames_of(system, lambda: ts.bound_potentials(system, np.float32, nblist_padding=0.1))
        rec = {"atoms": N}
        pots = {"Nonbonded": (P.Nonbonded(N, system.exclusion_idxs, system.scale_factors, system.beta, system.cutoff), system.nb_params)}
        if n_lig:
            lig = np.arange(N - n_lig, N, dtype=np.int32)
            pots["NonbondedInteractionGroup"] = (P.NonbondedInteractionGroup(N, lig, system.beta, system.cutoff), system.nb_params)
            state = ts.rbfe_shaped_state(system, n_lig)
            label = "SummedPotential(" + ", ".join(type(p).__name__ for p, _ in state) + ")"
            pots[label] = (P.SummedPotential([p for p, _ in state], [q for _, q in state]), np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1) for _, q in state]))
        for label, (pot, prm) in pots.items():
            prm = np.asarray(prm, dtype=np.float64)
            same = np.stack([prm] * n_params)
            rec[label] = {}
            for prec, tag in ((np.float32, "f32"), (np.float64, "f64")):
                unbound = pot.to_gpu(prec).unbound_impl
                r = {form: timed(unbound, xs, same, boxes, flags) for form, flags in forms.items()}
                if n_lig:
                    # five lambda windows: the ligand's charges and w differ, everything else is the state's
                    windows = np.stack([prm] * n_params)
                    flat_lig = None
                    if label.startswith("SummedPotential"):
                        sizes = [int(np.asarray(q).size) for _, q in state]
                        off = sum(sizes[:-1])  # the interaction group's parameters are the last block
                        flat_lig = off + 4 * (N - n_lig)
                    for k in range(n_params):
                        lam = 0.1 * k
                        view = windows[k].reshape(-1)[flat_lig:].reshape(-1, 4) if flat_lig is not None else windows[k].reshape(-1, 4)[N - n_lig :]
                        view[:, 0] *= 1.0 - 0.5 * lam
                        view[:, 3] = lam * system.cutoff
                    r["u_five_lambda_windows"] = timed(unbound, xs, windows, boxes, forms["u"])
                rec[label][tag] = r
        out["systems"][name] = rec
    # one execute() of every term of config 2, host to host (the reference's compare_forces call shape, tests/common.py:250-334)
    s2 = ts.small_solvated_ligand(lamb=0.3)
    single = {}
    for prec, tag in ((np.float32, "f32"), (np.float64, "f64")):
        impls = [(P.Nonbonded(s2.num_atoms, s2.exclusion_idxs, s2.scale_factors, s2.beta, s2.cutoff).to_gpu(prec).unbound_impl, s2.nb_params),
                 (P.HarmonicBond(s2.bond_idxs).to_gpu(prec).unbound_impl, s2.bond_params), (P.HarmonicAngle(s2.angle_idxs).to_gpu(prec).unbound_impl, s2.angle_params),
                 (P.PeriodicTorsion(s2.torsion_idxs).to_gpu(prec).unbound_impl, s2.torsion_params)]
        per = {}
        for impl, prm in impls:
            tms = []
            for rep in range(22):
                t0 = time.perf_counter()
                impl.execute(s2.coords, prm, s2.box, True, True, True)
                tms.append(time.perf_counter() - t0)
            per[type(impl).__name__] = 1e6 * float(np.mean(tms[2:]))
        single[tag] = {"us_per_execute": per, "us_all_four_terms": sum(per.values())}
    out["config2_single_execute"] = single
    emit_json(out)


# ---------------------------------------------------------------------------------------------------------------------
# --mode hrex
# ---------------------------------------------------------------------------------------------------------------------
def _runtime_env(name):
    """an environment variable as the C runtime sees it (the native library exports GPU_MAX_HW_QUEUES with setenv when it is loaded:
    os.environ, a copy made at interpreter start, does not show that)"""
    import ctypes

    try:
        libc = ctypes.CDLL(None)
        libc.getenv.restype = ctypes.c_char_p
        v = libc.getenv(name.encode())
        return v.decode() if v is not None else None
    except (OSError, AttributeError):  # pragma: no cover
        return os.environ.get(name)


def _group_threads(n_atoms):
    """the enqueueing host threads Context::multiple_steps_group uses for contexts of this size (csrc/integrator.hip)"""
    e = os.environ.get("TM_AMD_GROUP_THREADS")
    return max(1, int(e)) if e else (2 if n_atoms <= 5000 else 1)


def run_hrex(args, rank, local_rank, world, backend):
    from timemachine_amd import hrex, parallel

    n_states = args.windows or 24
    steps_per_frame = args.steps_per_frame
    co = None
    if not args.stub:
        from timemachine_amd import potentials as P
        from timemachine_amd import testsystems as ts
        from timemachine_amd.lib import LangevinIntegrator, MonteCarloBarostat, custom_ops as co

        if co.device_count() < 1:
            raise SystemExit("bench.py needs a GPU: timemachine_amd has no CPU fallback")
        co.set_device(0 if args.share_gpu else local_rank)
        system = ts.config5_complex_sized(0.0)
        N = system.num_atoms
        lig = np.arange(system.num_water_atoms, N)
        lambdas = np.linspace(0.0, 0.5, n_states)
        params_by_state = np.stack([system.nb_params] * n_states)
        for k, lam in enumerate(lambdas):  # windows differ only in the ligand's nonbonded parameters (4D decoupling + charge scaling)
            params_by_state[k][lig, 3] = lam * system.cutoff
            params_by_state[k][lig, 0] *= 1.0 - 0.5 * lam

        def make_bps(p):
            return [bp.to_gpu(p).bound_impl for bp in ts.bound_potentials(system, p, nblist_padding=args.padding)]

        # the reference's RBFE state composition (fe/system.py:133-146; testsystems.rbfe_shaped_state): windows differ in the interaction
        # group's parameters (ligand rows); the energy matrix evaluates the state's two tile producers -- host-host Nonbonded +
        # ligand-environment group, packed into one SummedPotential -- under the neighbouring windows' parameters
        n_lig = N - system.num_water_atoms
        rbfe_state = ts.rbfe_shaped_state(system, n_lig, nblist_padding=args.padding)
        group_params_by_state = np.stack([np.asarray(rbfe_state[7][1], dtype=np.float64)] * n_states)
        for k, lam in enumerate(lambdas):
            group_params_by_state[k][lig, 3] = lam * system.cutoff
            group_params_by_state[k][lig, 0] *= 1.0 - 0.5 * lam
        host_params_flat = np.asarray(rbfe_state[6][1], dtype=np.float64).reshape(-1)
        rbfe_matrix_params = np.stack([np.concatenate([host_params_flat, group_params_by_state[k].reshape(-1)]) for k in range(n_states)])

        def make_rbfe_bps(p):
            return [pot.bind(np.asarray(prm, dtype=np.float64)).to_gpu(p).bound_impl for pot, prm in rbfe_state]

        x0, v0 = equilibrate(co, LangevinIntegrator, system, make_bps, 99, args.equil_scale, np.float32)
    else:
        N = 31000

    def measure(precision_name, barostat_interval, n_frames, warm_frames, composition="single_nonbonded"):
        """one HREX measurement: this rank's resident replicas built afresh, warm-up frames, n_frames timed; -> (record fields, per_rank)
        composition: "single_nonbonded" (the benchmark states' shape: one all-atom Nonbonded) or "rbfe" (HostGuestSystem)"""
        dh = hrex.DistributedHREX(n_states, TEMPERATURE, max_delta_states=args.max_delta_states, world_size=world, rank=rank)
        mine = dh.local_replicas
        rbfe = composition == "rbfe" and not args.stub
        state_params = None
        if not args.stub:
            prec = np.float64 if precision_name == "f64" else np.float32
            if rbfe:
                unbound = P.SummedPotential([rbfe_state[6][0], rbfe_state[7][0]], [rbfe_state[6][1], rbfe_state[7][1]]).to_gpu(prec).unbound_impl
                state_params = rbfe_matrix_params
            else:
                unbound = P.Nonbonded(N, system.exclusion_idxs, system.scale_factors, system.beta, system.cutoff).to_gpu(prec).unbound_impl
                state_params = params_by_state
            ctxts, bound_nb = [], []
            for r in mine:
                bps = make_rbfe_bps(prec) if rbfe else make_bps(prec)
                bps[-1].set_params((group_params_by_state if rbfe else params_by_state)[r].reshape(-1))
                # the production shape (fe/rbfe.py:113-121,191-192; fe/free_energy.py:695-708): a barostat in every window's context
                movers = [MonteCarloBarostat(N, 1.0, TEMPERATURE, ts.molecule_groups(system), barostat_interval, 700 + r).impl(bps)] if barostat_interval > 0 else []
                ctxts.append(co.Context(x0, v0, system.box, LangevinIntegrator(TEMPERATURE, DT, FRICTION, system.masses, 500 + r).impl(), bps, movers=movers))
                bound_nb.append(bps[-1])
        else:
            ctxts = [StubContext(N) for _ in mine]
        timers = {"md": 0.0, "matrix": 0.0, "exchange": 0.0, "rebind": 0.0}

        def frame(it, timed):
            t0 = time.perf_counter()
            hrex.step_replicas(ctxts, steps_per_frame, group=args.replica_group)  # the rank's replicas, `group` at a time on one GPU
            device_sync(co)
            t1 = time.perf_counter()
            if args.stub:
                state = dh.state_of_replica()
                rows = np.full((len(mine), n_states), np.inf)
                for i, r in enumerate(mine):
                    lo, hi = max(0, state[r] - args.max_delta_states), min(n_states - 1, state[r] + args.max_delta_states)
                    rows[i, lo : hi + 1] = 0.1 * np.abs(np.arange(lo, hi + 1) - r)
            else:
                coords = np.stack([c.get_x_t() for c in ctxts])
                boxes = np.stack([c.get_box() for c in ctxts])  # (every window has its own box once a barostat is at work)
                rows = hrex.compute_potential_matrix(unbound, coords, boxes, state_params, dh.replica_idx_by_state, args.max_delta_states, replicas=mine)
            t2 = time.perf_counter()
            new_states = dh.exchange(rows, seed=1000 + it)  # collective: one all_gather + the identical swap chain everywhere
            t3 = time.perf_counter()
            if not args.stub:
                for i in range(len(mine)):
                    bound_nb[i].set_params((group_params_by_state if rbfe else params_by_state)[new_states[i]].reshape(-1))
            t4 = time.perf_counter()
            if timed:
                timers["md"] += t1 - t0
                timers["matrix"] += t2 - t1
                timers["exchange"] += t3 - t2
                timers["rebind"] += t4 - t3

        for it in range(warm_frames):
            frame(it, False)
        parallel.barrier()
        device_sync(co)
        t0 = time.perf_counter()
        c0 = time.process_time()  # user + system CPU time of this process, all threads (enqueue threads and the HIP runtime's included)
        for it in range(n_frames):
            frame(warm_frames + it, True)
        device_sync(co)
        cpu_s = time.process_time() - c0
        parallel.barrier()
        elapsed_rank = time.perf_counter() - t0
        elapsed = parallel.max_over_ranks(elapsed_rank)
        for c in ctxts:
            assert np.all(np.isfinite(c.get_x_t())), "trajectory diverged"
        t_max = {k: parallel.max_over_ranks(v) for k, v in timers.items()}
        md_steps = n_frames * steps_per_frame
        replica_steps = max(md_steps * max(len(mine), 1), 1)
        # who holds what: every rank's resident replicas (windows), its own MD time per frame, its host CPU, device and bus id
        per_rank = parallel.gather_objects({
            "rank": rank, "local_rank": local_rank, "resident_replicas": [int(r) for r in mine], "md_ms_per_frame": 1e3 * timers["md"] / n_frames,
            "ms_per_step": 1e3 * timers["md"] / replica_steps,
            "host_cpu_us_per_step": 1e6 * cpu_s / replica_steps, "host_cpu_load": cpu_s / max(elapsed_rank, 1e-9), "cpus": len(PINNED_CPUS) if PINNED_CPUS else None,
            "device": "stub" if args.stub else co.device_name(), "pci_bus_id": pci_bus_id(0 if args.share_gpu else local_rank, args.stub), "host": socket.gethostname(),
        })
        # every rank ran the swap chain for itself on the gathered matrix: the permutations must be the same everywhere
        chains = parallel.gather_objects([list(map(int, p)) for p in dh.replica_idx_by_state_by_iter] + [list(map(int, dh.replica_idx_by_state))])
        if rank != 0:
            return None, None
        chains_identical = all(c == chains[0] for c in chains)
        accepted = sum(a for it in dh.fraction_accepted_by_pair_by_iter[-n_frames:] for a, _ in it)
        proposed = sum(p for it in dh.fraction_accepted_by_pair_by_iter[-n_frames:] for _, p in it)
        fields = {
            "value": n_states * md_steps / elapsed * 86400.0 * DT * 1e-3,
            "steps": md_steps,
            "warmup": warm_frames * steps_per_frame,
            "ms_per_step": 1e3 * elapsed / md_steps,
            "dtype": precision_name,
            "barostat_interval": barostat_interval,
            "frames": n_frames,
            "frames_per_s": n_frames / elapsed,
            "per_frame_ms": {k: 1e3 * v / n_frames for k, v in t_max.items()},
            "exchange_latency_ms": 1e3 * (t_max["matrix"] + t_max["exchange"] + t_max["rebind"]) / n_frames,
            "swap_acceptance": accepted / max(proposed, 1),
            "value_per_gpu": n_states * md_steps / elapsed * 86400.0 * DT * 1e-3 / world,
            # what the host spends: process CPU time per replica-step (max over ranks) and CPUs busy over the whole job, to be held
            # against cpu_quota (the job's, taken before the ranks pinned themselves); how the grouped call was fed
            "host_cpu_us_per_step": max(r["host_cpu_us_per_step"] for r in per_rank),
            "host_cpu_load": sum(r["host_cpu_load"] for r in per_rank),
            "cpu_quota": JOB_CPU_QUOTA if JOB_CPU_QUOTA is not None else _cpu_quota(),
            "replica_group": args.replica_group,
            "enqueue_threads": _group_threads(N),
            "gpu_max_hw_queues": None if args.stub else _runtime_env("GPU_MAX_HW_QUEUES"),
            "resident_replicas_rank0": len(mine),
            "swap_chains_identical_across_ranks": chains_identical,
            "composition": composition,
        }
        if rbfe:
            nb0 = find_all_pairs(ctxts[0].get_potentials()) if ctxts else None
            if nb0 is not None:
                fields["merged_evaluations_replica0"] = nb0.get_merged_stats()[0]
            fields["energy_memo_matrix_potential"] = dict(zip(("evaluations", "all_pairs_launch_skipped"), find_all_pairs_of(unbound).get_memo_stats()))
            movers0 = ctxts[0].get_movers() if ctxts else []
            if movers0:
                fields["barostat_attempt_paths_replica0"] = dict(zip(("attempts", "on_current_list"), movers0[0].get_attempt_paths()))
        return fields, per_rank

    n_frames = max(args.steps // steps_per_frame, 1)
    warm_frames = max(args.warmup // steps_per_frame, 1)
    main_fields, per_rank = measure(args.precision, args.barostat_interval, n_frames, warm_frames)
    # the reference's PRODUCTION shape as a leg of the default line: f32 potentials, a Monte Carlo barostat every 25 steps in every
    # window (examples/run_rbfe_legs.py -> fe/rbfe.py:113-121,191-192; fe/free_energy.py:695-708), fewer frames
    production = production_single = None
    if args.precision == "f64" and args.barostat_interval == 0 and not args.no_npt:
        production, _ = measure("f32", 25, max(n_frames // 2, 1), 1, composition="rbfe")
        production_single, _ = measure("f32", 25, max(n_frames // 2, 1), 1)
    if rank != 0:
        return
    record = {
        "metric": "ns/day aggregate over all lambda windows, HREX (BASELINE config 5 shape)",
        "value": main_fields["value"],
        "unit": "ns/day",
        "n_gpus": world,
        "steps": main_fields["steps"],
        "warmup": main_fields["warmup"],
        "ms_per_step": main_fields["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": f"configs[4] (zero-based; BASELINE's fifth) shape: {n_states} lambda windows of a {N}-atom solvated-ligand state, round-robin over {world} GPU(s) "
            f"({main_fields['resident_replicas_rank0']} resident replicas on rank 0, stepped {args.replica_group} at a time), {steps_per_frame} steps per frame, neighbour exchange with max_delta_states {args.max_delta_states}, "
            f"{n_states}^3 swap attempts per frame; ms_per_step is wall time per MD step of ONE window slot (all windows of a rank run back to back)"
            + (f"; Monte Carlo barostat every {args.barostat_interval} steps in every window" if args.barostat_interval else ""),
            "atoms": N, "windows": n_states, "frames": main_fields["frames"], "nblist_padding": args.padding, "barostat_interval": args.barostat_interval,
        },
        "world_size": world,
        "backend": backend,
        "rccl_ranks": collective_ranks(backend),
        "per_rank": per_rank,
        "binding": "stub" if args.stub else co.BINDING,
        "device": "stub" if args.stub else co.device_name(),
    }
    for k in ("frames_per_s", "per_frame_ms", "exchange_latency_ms", "swap_acceptance", "value_per_gpu", "host_cpu_us_per_step", "host_cpu_load", "cpu_quota",
              "replica_group", "enqueue_threads", "gpu_max_hw_queues", "swap_chains_identical_across_ranks"):
        record[k] = main_fields[k]
    if args.share_gpu:
        record["share_gpu"] = True
        record["share_gpu_note"] = (f"REHEARSAL: all {world} ranks drive device 0 (gloo collectives); `value` is ONE GPU's aggregate over {world} concurrently "
                                    "working processes -- compare it with the one-process hrex line, not with an N-GPU figure")
    if production is not None:
        production["note"] = ("the reference's production shape (examples/run_rbfe_legs.py: fe/rbfe.py:113-121,191-192, fe/free_energy.py:695-708): every window a "
                              "HostGuestSystem (fe/system.py:133-146: bonded terms, chiral restraints, ligand-ligand precomputed pairs, host-host "
                              "Nonbonded(atom_idxs=host), ligand-environment NonbondedInteractionGroup), f32 potentials, a Monte Carlo barostat every 25 steps "
                              "in every window, replicas stepped together; same windows, half the frames; the energy matrix evaluates the two tile producers "
                              "of the state under the neighbouring windows' parameters (the host-host part once per frame: energy memo)")
        record["production_shape"] = production
    if production_single is not None:
        production_single["note"] = "the same ensemble on the BENCHMARK states' composition (one all-atom Nonbonded per window): rounds 4-5's production_shape"
        record["production_shape_single_nonbonded"] = production_single
    emit_json(record)


def main(argv=None):
    args = parse_args(argv)
    maybe_self_launch(args)
    protect_stdout()
    rank, local_rank, world, backend = init_distributed(args)
    global PINNED_CPUS, JOB_CPUS, JOB_CPU_QUOTA
    # the job-wide figures are taken BEFORE this rank cuts its affinity down to its own slice: host_cpu_load sums CPU time over
    # ALL ranks and belongs next to the whole job's quota (after pinning, _cpu_quota() is one rank's share)
    JOB_CPU_QUOTA = _cpu_quota()
    try:
        JOB_CPUS = sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        JOB_CPUS = None
    PINNED_CPUS = pin_rank_to_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    try:
        if args.mode == "potentials":
            if rank == 0:
                run_potentials(args)
        elif args.mode == "hrex":
            run_hrex(args, rank, local_rank, world, backend)
        else:
            run_md(args, rank, local_rank, world, backend)
    finally:
        if world > 1 or os.environ.get("TM_AMD_FORCE_COLLECTIVES"):
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    main()
