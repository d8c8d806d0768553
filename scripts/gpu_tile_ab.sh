#!/bin/bash
# A/B of tile-kernel library variants on one box: gpu_tile_ab.sh <lib.so | default | ENV=VAL,lib.so> ...
# (scripts/tile_ablate.py per variant, twice, alternating; results in gpurun_out/abl.log)
mkdir -p gpurun_out
: > gpurun_out/abl.log
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec##*,}
  envs=""
  [ "$lib" != "$spec" ] && envs=${spec%,*}
  (
    [ -n "$envs" ] && export ${envs//,/ }
    if [ "$lib" = "default" ]; then unset TM_AMD_LIB; else export TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/$lib; fi
    echo "[$spec]" >> gpurun_out/abl.log
    timeout 300 python scripts/tile_ablate.py 2>&1 | tail -3 >> gpurun_out/abl.log
  )
done
done
cat gpurun_out/abl.log
