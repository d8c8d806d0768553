#!/bin/bash
# A/B bench of library variants: gpu_ab.sh <steps> <lib-or-"default"> [...]; one JSON line per variant in gpurun_out/ab.log
set -u
steps=$1; shift
mkdir -p gpurun_out
: > gpurun_out/ab.log
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset TM_AMD_LIB; else export TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/$lib; fi
  echo "== $lib" >> gpurun_out/ab.log
  timeout 600 python bench.py --steps $steps --warmup 200 --no-cpu-baseline ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32')}, 'tile_ms', d['roofline']['kernel_ms'])" >> gpurun_out/ab.log 2>&1
done
cat gpurun_out/ab.log
