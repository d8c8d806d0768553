#!/bin/bash
# A/B of a process-wide switch read from the environment: scripts/gpu_env_ab.sh <VAR> <steps> [bench args]; runs VAR=0,1,0,1
# -> gpurun_out/env_ab.log
set -u
var=$1; steps=$2; shift 2
mkdir -p gpurun_out
: > gpurun_out/env_ab.log
for on in 0 1 0 1; do
  echo "== $var=$on" >> gpurun_out/env_ab.log
  env $var=$on timeout 300 python bench.py --steps $steps --warmup 200 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32','ns_day_npt','ns_day_npt_f32','ns_day_rc1.0_f32')}, 'tile_ms', d['roofline']['kernel_ms'])" >> gpurun_out/env_ab.log 2>&1
done
cat gpurun_out/env_ab.log
