"""Copy what scripts/gpu_round_artifacts.sh <tag> left under gpurun_out/ into profiles/ under round-style names.
usage: python scripts/collect_profiles.py <artifact tag, e.g. r02v3> <profile prefix, e.g. r02_v3> "<one-line build description>" """
import json, os, shutil, sys
tag, prefix, desc = sys.argv[1], sys.argv[2], sys.argv[3]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P, A = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles"), os.path.join(R, "gpurun_out", "art_" + tag)
for prec in ("f64", "f32"):
    shutil.copy(os.path.join(G, f"prof_{tag}_{prec}", "p_kernel_stats.csv"), os.path.join(P, f"{prefix}_kernel_stats_{prec}.csv"))
    hdr = (f"# rocprofv3 --kernel-trace --stats of: python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-npt --no-rc10 --no-rbfe-shape --profile-steps 0{' --precision f32' if prec == 'f32' else ''}"
           f"   (scripts/gpu_profile.sh; {desc})\n# per-step table = the last 350 timed MD steps of the kernel trace; times in us\n")
    open(os.path.join(P, f"{prefix}_per_step_{prec}.txt"), "w").write(hdr + open(os.path.join(A, f"profile_{prec}.txt")).read())
    lines = [f"# rocprofv3 PMC summary, {desc}",
             f"# command per pass: rocprofv3 --pmc <counters> --output-format csv -- python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-npt --no-rc10 --no-rbfe-shape --profile-steps 0 --equil-scale 0.2 --equil-precision {prec} --precision {prec}   (scripts/gpu_pmc.sh; one pass per counter group, counters only)",
             "# per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KB (raw; the HBM guide's gfx950 correction doubles FETCH_SIZE); SQ_* cycle counters summed over waves (quad-cycle units); SQ_INSTS_* / SQ_WAVES are counts"]
    for name in ("sq", "sq2", "fetch", "write"):
        lines.append(f"== pass {name}")
        lines += [l.rstrip() for l in open(os.path.join(G, f"pmc_md_{prec}", name, f"{name}_summary.txt"))]
    open(os.path.join(P, f"{prefix}_pmc_md_{prec}.txt"), "w").write("\n".join(lines) + "\n")
traffic = {}
for prec in ("f64", "f32"):
    d = json.load(open(os.path.join(G, f"pmc_md_{prec}", "pmc_traffic.json")))[prec]
    d["source"] = f"profiles/{prefix}_pmc_md_{prec}.txt"
    traffic[prec] = d
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(A, "bench_md.json"), os.path.join(P, f"{prefix}_bench.json"))
shutil.copy(os.path.join(A, "bench_hrex.json"), os.path.join(P, f"{prefix}_bench_hrex.json"))
d = json.load(open(os.path.join(P, f"{prefix}_bench.json")))
print({k: d[k] for k in ("value", "ms_per_step", "ns_day_f32")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline_valu"]["frac"], d["roofline"]["traffic_source"][:40])
