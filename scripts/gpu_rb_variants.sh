#!/bin/bash
# A/B of row-block kernel shapes: gpu_rb_variants.sh <variant-tag> ... ("default" = the product library)
set -u
mkdir -p gpurun_out
L=gpurun_out/rb_variants.log
: > $L
for tag in "$@"; do
  if [ "$tag" = "default" ]; then unset TM_AMD_LIB; else export TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_$tag.so; fi
  echo "== $tag" >> $L
  TM_AMD_ROWBLOCK_MIN_K=${RB_MIN_K:-8192} timeout 200 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-npt --no-rc10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32')}, 'tile_ms', d['roofline']['kernel_ms'])" >> $L 2>&1
done
if [ -n "${TIMING_TAG:-}" ]; then
  echo "== timing $TIMING_TAG" >> $L
  TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_$TIMING_TAG.so timeout 200 python scripts/rb_timing.py >> $L 2>&1
fi
cat $L
