#!/bin/bash
# round 6, verification of HEAD's library (the carrier take-over fix came after profiles/r06_v4): smoke, the GPU suite, the default
# bench line, kernel stats in both precisions, the RBFE-composition traces, and the interleavings campaign over fresh seeds.
# usage: scripts/gpu_r6_v5.sh <tag> [campaign seeds] [first seed]
set -u
TAG=${1:-r06v5}; NSEED=${2:-40}; SEED0=${3:-5000}
R=$GRAFT_REPO_ROOT
A=gpurun_out/art_$TAG
mkdir -p $A
export PYTHONPATH=$R TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > $A/smoke.txt 2>&1; echo "smoke exit $?"; tail -3 $A/smoke.txt
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 ) > $A/gpu_tests.txt 2>&1; tail -2 $A/gpu_tests.txt
echo "== profile f64"; bash scripts/gpu_profile.sh ${TAG}_f64 > $A/profile_f64.txt 2>&1; tail -12 $A/profile_f64.txt
echo "== profile f32"; bash scripts/gpu_profile.sh ${TAG}_f32 --precision f32 > $A/profile_f32.txt 2>&1; tail -8 $A/profile_f32.txt
echo "== bench default"; timeout 900 python bench.py > $A/bench_md.json 2> $A/bench_md.err; echo "exit $?"; tail -c 600 $A/bench_md.json
for a in "config5 f64" "config5 f64 --no-merge"; do
  t=$(echo $a | tr ' ' '_' | tr -d '-'); bash scripts/gpu_profile_rbfe.sh ${TAG}_$t $a > $A/per_step_rbfe_$t.txt 2>&1
done
( timeout 1500 python scripts/fuzz_campaign_long.py $NSEED $SEED0 2>&1 | grep -v amdgpu.ids ) > $A/fuzz_long.txt 2>&1; tail -5 $A/fuzz_long.txt
( timeout 900 python scripts/fuzz_campaign_config5.py 150 7000 600 2>&1 | grep -v amdgpu.ids ) > $A/fuzz_config5.txt 2>&1; tail -3 $A/fuzz_config5.txt
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; du -sh gpurun_out
