#!/bin/bash
# round 5, first GPU call: the whole GPU suite (new grouped-listed tests, variant-library row-block test), the tile-kernel
# experiment variants A/B on a fixed frame, a baseline bench line
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r5a_tests.log 2>&1
bash scripts/gpu_tile_ab.sh default libtimemachine_amd_estrin.so libtimemachine_amd_ljearly.so libtimemachine_amd_splitwait.so libtimemachine_amd_combo.so \
   libtimemachine_amd_prio1.so libtimemachine_amd_prio3.so libtimemachine_amd_abl10.so libtimemachine_amd_abl11.so libtimemachine_amd_abl12.so libtimemachine_amd_abl13.so > /dev/null 2>&1
cp gpurun_out/abl.log gpurun_out/r5a_abl.log
timeout 400 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
echo "== tests"; cat gpurun_out/r5a_tests.log
echo "== A/B"; cat gpurun_out/r5a_abl.log
echo "== bench"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5a_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32','npt','rc1.0_f32','rc1.0_f64')})
print('kernels', [(k['name'], round(k['us_per_step'],2), round(k['share_of_step'],3)) for k in d.get('kernels',[])])
print('replicas', d.get('replicas_per_gpu'))
PY
