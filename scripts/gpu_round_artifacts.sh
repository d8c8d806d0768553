#!/bin/bash
# One gpurun call that produces everything profiles/ holds for a round: guard check, kernel stats (f64, f32), PMC passes
# (f64, f32), the default bench line (with CPU baselines) and the hrex bench line.  usage: scripts/gpu_round_artifacts.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/art_$TAG
A=$ROOT/gpurun_out/art_$TAG
echo "== guard"; TM_AMD_LIB=$ROOT/timemachine_amd/csrc/libtimemachine_amd_guard.so timeout 400 python scripts/guard_check.py > $A/guard.log 2>&1; echo "guard exit $?"; tail -8 $A/guard.log
echo "== profile f64"; bash scripts/gpu_profile.sh ${TAG}_f64 > $A/profile_f64.txt 2>&1; tail -12 $A/profile_f64.txt
echo "== profile f32"; bash scripts/gpu_profile.sh ${TAG}_f32 --precision f32 > $A/profile_f32.txt 2>&1; tail -8 $A/profile_f32.txt
echo "== pmc f64"; PMC_PREC=f64 bash scripts/gpu_pmc.sh > $A/pmc_f64.txt 2>&1; grep -E "pmc pass|traffic|tiles<" $A/pmc_f64.txt | cut -c1-400
echo "== pmc f32"; PMC_PREC=f32 bash scripts/gpu_pmc.sh > $A/pmc_f32.txt 2>&1; grep -E "pmc pass|traffic" $A/pmc_f32.txt | cut -c1-300
# the bench line quotes the counters of THIS build: put the fresh PMC record where bench.py reads it before the bench runs
python - <<'PY'
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = {}
for prec in ("f64", "f32"):
    p = os.path.join(root, "gpurun_out", f"pmc_md_{prec}", "pmc_traffic.json")
    if os.path.exists(p):
        out.update(json.load(open(p)))
if out:
    json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
PY
echo "== bench default"; timeout 900 python bench.py > $A/bench_md.json 2> $A/bench_md.err; echo "exit $?"; tail -c 3000 $A/bench_md.json
echo "== bench hrex"; timeout 600 python bench.py --mode hrex > $A/bench_hrex.json 2> $A/bench_hrex.err; echo "exit $?"; tail -c 1500 $A/bench_hrex.json
du -sh $ROOT/gpurun_out
