#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
for own in 0 1; do
  if [ $own = 1 ]; then export TM_AMD_CONTEXT_STREAM=1; else unset TM_AMD_CONTEXT_STREAM; fi
  timeout 600 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-rc10 --no-npt > gpurun_out/r5i_bench_$own.json 2> gpurun_out/r5i_bench_$own.err
done
python - <<'PY'
import json
for own in (0,1):
    d=json.loads(open(f'gpurun_out/r5i_bench_{own}.json').read().strip().splitlines()[-1])
    print('own stream' if own else 'null stream', {k:d.get(k) for k in ('value','ms_per_step','host_ms_per_step','host_cpu_us_per_step','host_cpu_load','ns_day_f32')})
PY
