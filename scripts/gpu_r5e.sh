#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( scripts/microbench/persistent_step ) > gpurun_out/r5e_persistent.log 2>&1
( timeout 300 python scripts/tile_ablate.py 2>&1 | tail -2 ) > gpurun_out/r5e_tile.log 2>&1
( TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_tbatch.so timeout 300 python scripts/batch_timeline.py 2>&1 | tail -12 ) > gpurun_out/r5e_timeline.log 2>&1
( env -u GPU_MAX_HW_QUEUES TM_AMD_BINDING=ctypes GROUP_COUNTS=1,4 timeout 300 python scripts/group_bench.py f64 1500 2>&1 | tail -4 ) > gpurun_out/r5e_group_ctypes.log 2>&1
( timeout 300 python scripts/npt_bench.py f64 25 3000 2>&1 | tail -2 ) > gpurun_out/r5e_npt.log 2>&1
bash scripts/gpu_npt_trace.sh f64 > gpurun_out/r5e_trace.log 2>&1
for f in persistent tile timeline group_ctypes npt trace; do echo "== $f"; cat gpurun_out/r5e_$f.log; done
