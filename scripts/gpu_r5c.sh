#!/bin/bash
# round 5: NPT attempt variants -- dual launch (wide shape), two launches, two launches with the wide energy kernel
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_barostat_cases.py -m gpu -x -q -p no:cacheprovider -k "current_list" 2>&1 | tail -5 ) > gpurun_out/r5c_tests.log 2>&1
for rep in 1 2; do
( timeout 300 python scripts/npt_bench.py f64 25 3000 2>&1 | tail -1 ) >> gpurun_out/r5c_dual.log 2>&1
( TM_AMD_BAROSTAT_TWO_LAUNCHES=1 timeout 300 python scripts/npt_bench.py f64 25 3000 2>&1 | tail -1 ) >> gpurun_out/r5c_two.log 2>&1
( TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_ewide.so TM_AMD_BAROSTAT_TWO_LAUNCHES=1 timeout 300 python scripts/npt_bench.py f64 25 3000 2>&1 | tail -1 ) >> gpurun_out/r5c_two_ewide.log 2>&1
done
bash scripts/gpu_npt_trace.sh f64 > gpurun_out/r5c_trace.log 2>&1
for f in tests dual two two_ewide trace; do echo "== $f"; cat gpurun_out/r5c_$f.log; done
