#!/bin/bash
# round 6 artifacts: everything profiles/r06_<v>_* holds, on the final library (one gpurun call).  usage: scripts/gpu_r6_final.sh <tag>
set -u
TAG=${1:-r06v2}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$R
bash scripts/gpu_round_artifacts.sh $TAG > gpurun_out/art_$TAG.log 2>&1
A=gpurun_out/art_$TAG
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) > $A/gpu_tests.txt 2>&1
( TM_AMD_LIB=$R/timemachine_amd/csrc/libtimemachine_amd_guard.so timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 ) > $A/gpu_tests_guard.txt 2>&1
bash scripts/gpu_npt_trace.sh f64 > $A/npt_trace_f64.txt 2>&1
# the reference's RBFE composition: per-step traces (merged carrier, producers not merged, one all-atom Nonbonded)
for a in "config5 f64" "config5 f64 --no-merge" "config5 f64 --single" "config4 f32" "config4 f32 --no-merge" "config5 f32"; do
  t=$(echo $a | tr ' ' '_' | tr -d '-'); bash scripts/gpu_profile_rbfe.sh ${TAG}_$t $a > $A/per_step_rbfe_$t.txt 2>&1
done
# potential-evaluation throughput (benchmark_potential) + its kernel table; first / further parameter sets
timeout 900 python bench.py --mode potentials > $A/bench_potentials.json 2> $A/bench_potentials.err
bash scripts/gpu_stats_cmd.sh ${TAG}_pot 60 python $R/bench.py --mode potentials --systems dhfr > $A/potentials_kernel_stats_dhfr.txt 2>&1
bash scripts/gpu_stats_cmd.sh ${TAG}_pot5 60 python $R/bench.py --mode potentials --systems config5 > $A/potentials_kernel_stats_config5.txt 2>&1
( for a in "dhfr f64 same" "dhfr f32 same" "config5 f64 same" "config5 f64 windows" "config5 f32 windows"; do python scripts/further_sets_probe.py $a 2>&1 | grep -v amdgpu.ids; done
  echo "-- the same with the same-frame hint and the energy memo switched off"
  for a in "dhfr f64 same" "config5 f64 windows"; do TM_AMD_NO_ENERGY_MEMO=1 python scripts/further_sets_probe.py $a 2>&1 | grep -v amdgpu.ids; done ) > $A/further_sets.txt 2>&1
( for pr in f64 f32; do bash scripts/gpu_stats_cmd.sh ${TAG}_form_$pr 8 python $R/scripts/pp_launch_probe.py $pr | cut -c1-220; grep "device us" gpurun_out/prof_${TAG}_form_$pr/run.log; done ) > $A/tile_launch_by_form.txt 2>&1
( timeout 600 python scripts/matrix_probe.py 2>&1 | grep -v amdgpu.ids ) > $A/matrix_probe.txt 2>&1
# the 8-rank launch rehearsed on one GPU
timeout 900 python bench.py --gpus 8 --share-gpu > $A/bench_share_gpu_md.json 2> $A/bench_share_gpu_md.err
timeout 900 python bench.py --gpus 8 --share-gpu --mode hrex > $A/bench_share_gpu_hrex.json 2> $A/bench_share_gpu_hrex.err
( timeout 1500 python scripts/soak_rbfe.py 200000 2>&1 | grep -v amdgpu.ids ) > $A/soak_rbfe.txt 2>&1
# random interleavings (every fast path on = off) and random systems against the oracle, over many seeds
( timeout 900 python scripts/fuzz_campaign.py 2>&1 | grep -v amdgpu.ids; timeout 900 python scripts/fuzz_parity.py 100 300 2>&1 | grep -v amdgpu.ids
  echo "-- scripts/fuzz_campaign_long.py 100 9000"; timeout 900 python scripts/fuzz_campaign_long.py 100 9000 2>&1 | grep -v amdgpu.ids
  echo "-- scripts/fuzz_campaign_config5.py 300 9000 (single windows at config-5 size)"; timeout 900 python scripts/fuzz_campaign_config5.py 300 9000 600 2>&1 | grep -v amdgpu.ids ) > $A/fuzz.txt 2>&1
( bash scripts/gpu_nbl_probe.sh f64 product; bash scripts/gpu_nbl_probe.sh f32 product ) > $A/nbl_probe.txt 2>&1
tail -30 gpurun_out/art_$TAG.log
for f in gpu_tests gpu_tests_guard npt_trace_f64 further_sets matrix_probe soak_rbfe fuzz nbl_probe; do echo "== $f"; tail -12 $A/$f.txt | cut -c1-300; done
