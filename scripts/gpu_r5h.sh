#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) > gpurun_out/r5h_tests.log 2>&1
for spin in 0 1; do
  if [ $spin = 1 ]; then export TM_AMD_SPIN_WAIT=1; else unset TM_AMD_SPIN_WAIT; fi
  timeout 600 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-rc10 > gpurun_out/r5h_bench_$spin.json 2> gpurun_out/r5h_bench_$spin.err
done
unset TM_AMD_SPIN_WAIT
timeout 900 python bench.py --mode hrex --steps 1200 --warmup 400 > gpurun_out/r5h_hrex.json 2> gpurun_out/r5h_hrex.err
echo "== tests"; cat gpurun_out/r5h_tests.log
python - <<'PY'
import json
for spin in (0,1):
    d=json.loads(open(f'gpurun_out/r5h_bench_{spin}.json').read().strip().splitlines()[-1])
    print('spin' if spin else 'sleep-poll', {k:d.get(k) for k in ('value','ms_per_step','host_ms_per_step','host_cpu_us_per_step','host_cpu_load')}, 'npt', d['npt']['ns_day'], d['npt']['ratio_to_nvt_at_npt_box'])
    r=d.get('replicas_per_gpu',{})
    print('   replicas', {k:(round(v['aggregate_ns_day']), round(v['us_per_replica_step'],1), round(v['host_cpu_load'],2)) for k,v in r.items() if isinstance(v,dict)})
h=json.loads(open('gpurun_out/r5h_hrex.json').read().strip().splitlines()[-1])
print('hrex', {k:h.get(k) for k in ('value','host_cpu_us_per_step','host_cpu_load')}, 'production', {k:h['production_shape'].get(k) for k in ('value','host_cpu_us_per_step','host_cpu_load')})
PY
