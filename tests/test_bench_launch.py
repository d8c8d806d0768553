"""CPU tests of bench.py's launch / collective / report plumbing with a stubbed step (--stub: a stand-in Context that
sleeps; everything else -- self-launch of N ranks through torch.distributed.run, gloo process group, barriers, max over
ranks, the gather, HREX exchange, the JSON line -- is the code the GPU run uses over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--stub", *flags], capture_output=True, text=True, timeout=300, env=e, cwd=REPO)
    return r


def _line(r):
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    for k in CONTRACT:
        assert k in out, k
    assert "workload" in out["config"] and "model" not in out["config"]
    return out


def test_bench_single_rank_stub():
    out = _line(_run("--steps", "20", "--warmup", "5"))
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["value"] > 0 and out["host_ms_per_step"] >= out["ms_per_step"] * 0.5


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it becomes a 2-rank job (gloo here, RCCL on the GPU box)."""
    out = _line(_run("--gpus", "2", "--steps", "40", "--warmup", "10"))
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert out["config"]["replicas"] == 2 and out["scaling"] == "weak"
    assert out["mbar_gather_ok"] is True and out["mbar_gather_ms"] > 0
    single = _line(_run("--steps", "40", "--warmup", "10"))
    assert out["value"] > 1.2 * single["value"]  # aggregate over both ranks, not per rank
    # the N > 1 record explains itself: the per-GPU figure, who ran what, and (on an "nccl" group only) the RCCL rank count
    assert out["value_per_gpu"] == pytest.approx(out["value"] / 2)
    assert out["rccl_ranks"] is None  # gloo here; dist.get_world_size() of the nccl group on the GPU box
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and [r["windows"] for r in out["per_rank"]] == [[0], [1]]
    for r in out["per_rank"]:
        assert r["ms_per_step"] > 0 and r["device"] == "stub" and "pci_bus_id" in r and r["host"]
    assert max(r["ms_per_step"] for r in out["per_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-6)  # value = max over ranks
    assert single["value_per_gpu"] == pytest.approx(single["value"]) and len(single["per_rank"]) == 1


def test_bench_refuses_a_mismatched_launcher():
    r = _run("--gpus", "4", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stdout + r.stderr)


def test_bench_hrex_mode_two_ranks():
    out = _line(_run("--gpus", "2", "--mode", "hrex", "--windows", "6", "--steps", "800", "--warmup", "400"))
    assert out["n_gpus"] == 2 and out["config"]["windows"] == 6 and out["config"]["frames"] == 2
    assert out["scaling"] == "strong" and out["frames_per_s"] > 0 and out["exchange_latency_ms"] > 0
    assert 0.0 < out["swap_acceptance"] <= 1.0
    assert set(out["per_frame_ms"]) == {"md", "matrix", "exchange", "rebind"}
    # resident replicas per rank: round-robin placement (parallel.windows_for_rank), every window exactly once
    assert [r["resident_replicas"] for r in out["per_rank"]] == [[0, 2, 4], [1, 3, 5]]
    assert out["value_per_gpu"] == pytest.approx(out["value"] / 2) and out["rccl_ranks"] is None
    # the host in the record: CPU time per replica-step, CPUs busy over the job against the job's quota, how the grouped call was fed
    assert out["host_cpu_us_per_step"] > 0 and out["host_cpu_load"] > 0 and out["cpu_quota"] >= 1
    assert out["replica_group"] == 4 and out["enqueue_threads"] == 1 and "gpu_max_hw_queues" in out
    for p in out["per_rank"]:
        assert p["host_cpu_us_per_step"] > 0 and p["host_cpu_load"] > 0
    # the production-shape leg of the default line (f32 potentials, barostat every 25 steps in every window)
    prod = out["production_shape"]
    assert prod["dtype"] == "f32" and prod["barostat_interval"] == 25 and prod["value"] > 0 and prod["frames"] == 1
    assert prod["host_cpu_us_per_step"] > 0 and "fe/rbfe.py" in prod["note"]
    assert out["config"]["barostat_interval"] == 0


def test_bench_hrex_mode_with_a_barostat_is_one_line_without_the_extra_leg():
    out = _line(_run("--gpus", "1", "--mode", "hrex", "--windows", "4", "--steps", "400", "--warmup", "400", "--precision", "f32", "--barostat-interval", "25"))
    assert out["dtype"] == "f32" and out["config"]["barostat_interval"] == 25 and "production_shape" not in out


def test_cpu_baseline_times_whole_force_evaluations():
    """bench.py's config-3 CPU baseline (the oracle's dense restatement on the host cores): by default every row slab of the
    pair matrix, i.e. one whole force evaluation per timed pass and nothing extrapolated; a slab sample is scaled by pair count.
    Both must describe themselves, agree on what a pass covers, and keep the process's heap settings harmless."""
    import numpy as np

    import bench
    from timemachine_amd import testsystems as ts

    s = ts.build_water_box(300, 3.0)
    whole = bench.cpu_baseline(s, s.coords, 1.2, reps=2)
    assert whole["kind"] == "port" and whole["unit"] == "ns/day" and whole["cores"] >= 1
    assert "WHOLE i<j pair matrix" in whole["sample"] and "nothing extrapolated" in whole["sample"]
    assert len(whole["estimates_s"]) == 2 and all(e > 0 for e in whole["estimates_s"])
    assert np.isclose(whole["value"], 86400.0 * bench.DT * 1e-3 / whole["seconds_per_force_eval"])
    assert whole["spread_rel"] >= 0.0
    sampled = bench.cpu_baseline(s, s.coords, 1.2, reps=1, slabs_per_rep=3, rows_per_slab=32)
    assert "scaled to the full matrix by pair count" in sampled["sample"] and sampled["estimates_s"][0] > 0


def test_bench_eight_ranks_on_a_small_cpu_set():
    """The driver's largest job: `--gpus 8` = eight ranks on one host.  Its launch / barrier / max-over-ranks / gather path at the
    REAL rank count, with the whole job confined to as few CPUs as the GPU boxes' cgroup grants in the worst case seen (this
    machine's share, at most 16): every rank pins itself to its own slice of the allowed CPUs (bench.pin_rank_to_cpus) and reports
    what it spends per step, so that an N = 8 record shows whether the host could have been the limit."""
    allowed = sorted(os.sched_getaffinity(0))[:16]
    prefix = ["taskset", "-c", ",".join(str(c) for c in allowed)] if len(allowed) >= 8 else []
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)
    r = subprocess.run(prefix + [sys.executable, os.path.join(REPO, "bench.py"), "--stub", "--gpus", "8", "--steps", "40", "--warmup", "10"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=REPO)
    out = _line(r)
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["backend"] == "gloo" and out["config"]["replicas"] == 8
    assert [p["rank"] for p in out["per_rank"]] == list(range(8)) and [p["windows"] for p in out["per_rank"]] == [[k] for k in range(8)]
    assert out["mbar_gather_ok"] is True
    assert out["value_per_gpu"] == pytest.approx(out["value"] / 8)
    # host-side cost of a step, per rank and for the job, against what the job may use
    assert out["cpu_quota"] >= 1 and out["host_cpu_us_per_step"] > 0 and out["host_cpu_load"] > 0
    for p in out["per_rank"]:
        assert p["host_cpu_us_per_step"] > 0
        if prefix and len(allowed) >= 16:
            assert p["cpus"] == len(allowed) // 8  # each rank has its own slice


def test_pin_rank_to_cpus_slices_the_allowed_set():
    import bench

    before = sorted(os.sched_getaffinity(0))
    try:
        if len(before) >= 4:
            mine = bench.pin_rank_to_cpus(1, 2)
            assert mine == before[len(before) // 2:2 * (len(before) // 2)] and sorted(os.sched_getaffinity(0)) == mine
            os.sched_setaffinity(0, before)
        assert bench.pin_rank_to_cpus(0, 1) == before  # a single rank keeps everything
    finally:
        os.sched_setaffinity(0, before)


def test_bench_share_gpu_rehearsal_of_the_n_rank_launch():
    """`--gpus N --share-gpu`: N ranks that all drive device 0 over gloo -- the one-GPU rehearsal of the 8-GPU job (rank launch, CPU
    pinning, N device contexts, the collectives, grouped stepping).  The record says what it is, the hrex line vouches that every rank
    ran the same swap chain, and RCCL is refused (it cannot put two ranks on one device)."""
    md = _line(_run("--gpus", "2", "--share-gpu", "--steps", "40", "--warmup", "10"))
    assert md["share_gpu"] is True and md["backend"] == "gloo" and "REHEARSAL" in md["share_gpu_note"] and md["mbar_gather_ok"] is True
    hx = _line(_run("--gpus", "2", "--share-gpu", "--mode", "hrex", "--windows", "6", "--steps", "800", "--warmup", "400"))
    assert hx["share_gpu"] is True and hx["swap_chains_identical_across_ranks"] is True
    assert [r["resident_replicas"] for r in hx["per_rank"]] == [[0, 2, 4], [1, 3, 5]]
    plain = _line(_run("--gpus", "2", "--mode", "hrex", "--windows", "6", "--steps", "400", "--warmup", "400"))
    assert "share_gpu" not in plain and plain["swap_chains_identical_across_ranks"] is True
    r = _run("--gpus", "2", "--share-gpu", "--backend", "nccl", "--steps", "10", "--warmup", "5")
    assert r.returncode != 0
