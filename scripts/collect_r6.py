"""Copy what scripts/gpu_r6_final.sh <tag> left under gpurun_out/art_<tag> into profiles/<prefix>_* (after scripts/collect_profiles.py did the
bench lines, kernel stats and PMC passes).   usage: python scripts/collect_r6.py <tag> <prefix>"""
import os, shutil, sys
tag, prefix = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, P = os.path.join(R, "gpurun_out", "art_" + tag), os.path.join(R, "profiles")
names = {
    "guard.log": "guard_check.txt", "npt_trace_f64.txt": "npt_attempt_trace_f64.txt", "bench_potentials.json": "bench_potentials.json",
    "bench_share_gpu_md.json": "bench_share_gpu_8ranks_md.json", "bench_share_gpu_hrex.json": "bench_share_gpu_8ranks_hrex.json",
    "further_sets.txt": "further_sets.txt", "matrix_probe.txt": "matrix_probe.txt", "soak_rbfe.txt": "soak_rbfe.txt",
    "tile_launch_by_form.txt": "tile_launch_by_form.txt", "potentials_kernel_stats_dhfr.txt": "potentials_kernel_stats_dhfr.txt",
    "potentials_kernel_stats_config5.txt": "potentials_kernel_stats_config5.txt", "fuzz.txt": "fuzz_campaigns.txt", "nbl_probe.txt": "nbl_probe.txt",
}
for f in sorted(os.listdir(A)):
    if f.startswith("per_step_rbfe_"):
        names[f] = f
for src, dst in names.items():
    p = os.path.join(A, src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(P, f"{prefix}_{dst}"))
        print("copied", src, "->", f"{prefix}_{dst}")
    else:
        print("MISSING", src)
with open(os.path.join(P, f"{prefix}_gpu_tests.txt"), "w") as out:
    for f, title in (("gpu_tests.txt", "python -m pytest tests -m gpu -x -q   (product library)"), ("gpu_tests_guard.txt", "the same under the guard-zone build (TM_AMD_LIB=libtimemachine_amd_guard.so)")):
        p = os.path.join(A, f)
        out.write(f"# {title}\n" + (open(p).read() if os.path.exists(p) else "missing\n"))
