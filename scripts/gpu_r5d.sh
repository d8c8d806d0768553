#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/r5d_tests.log 2>&1
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err
timeout 900 python bench.py --mode hrex --steps 1200 --warmup 400 > gpurun_out/r5d_hrex.json 2> gpurun_out/r5d_hrex.err
echo "== tests"; cat gpurun_out/r5d_tests.log
echo "== bench"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5d_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32','rc1.0_f32','rc1.0_f64')})
print('npt', d.get('npt'))
print('kernels', [(k['name'], round(k['us_per_step'],2), round(k['share_of_step'],3)) for k in d.get('kernels',[])])
r=d.get('replicas_per_gpu',{})
print('replicas', {k:(round(v['aggregate_ns_day']), round(v['us_per_replica_step'],1), round(v['host_cpu_load'],2), v['gpu_max_hw_queues']) for k,v in r.items() if isinstance(v,dict)})
h=json.loads(open('gpurun_out/r5d_hrex.json').read().strip().splitlines()[-1])
print('hrex', {k:h.get(k) for k in ('value','per_frame_ms','host_cpu_us_per_step','host_cpu_load','cpu_quota','enqueue_threads','gpu_max_hw_queues')})
print('hrex production', {k:h['production_shape'].get(k) for k in ('value','per_frame_ms','host_cpu_us_per_step','host_cpu_load')})
PY
tail -3 gpurun_out/r5d_bench.err gpurun_out/r5d_hrex.err
