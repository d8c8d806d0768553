#!/bin/bash
# fixed-frame tile-kernel times of library variants: gpu_rb_frames.sh <tag> ...   (TIMING_TAG=<tag>: phase breakdown too)
set -u
mkdir -p gpurun_out
L=gpurun_out/rb_frames.log
: > $L
timeout 200 python scripts/rb_frame_bench.py >> $L 2>&1
for tag in "$@"; do
  TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_$tag.so RB_MIN_KS=0 timeout 200 python scripts/rb_frame_bench.py 2>&1 | tail -3 >> $L
done
if [ -n "${TIMING_TAG:-}" ]; then
  echo "== timing $TIMING_TAG" >> $L
  TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_$TIMING_TAG.so timeout 200 python scripts/rb_timing.py >> $L 2>&1
fi
cat $L
