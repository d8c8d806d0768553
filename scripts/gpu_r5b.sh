#!/bin/bash
# round 5: the barostat's fast path -- tests, NPT rate (dual launch / two launches / reference-shaped), the attempt's launch trace
set -u
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_barostat_cases.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "barostat or npt or current_list or group_stepping" 2>&1 | tail -25 ) > gpurun_out/r5b_tests.log 2>&1
( timeout 300 python scripts/npt_bench.py f64 25 2000 2>&1 | tail -3 ) > gpurun_out/r5b_npt_fast.log 2>&1
( TM_AMD_BAROSTAT_TWO_LAUNCHES=1 timeout 300 python scripts/npt_bench.py f64 25 2000 2>&1 | tail -3 ) > gpurun_out/r5b_npt_two.log 2>&1
( TM_AMD_BAROSTAT_SLOW_PATH=1 timeout 300 python scripts/npt_bench.py f64 25 2000 2>&1 | tail -3 ) > gpurun_out/r5b_npt_slow.log 2>&1
( timeout 300 python scripts/npt_bench.py f32 25 2000 2>&1 | tail -3 ) > gpurun_out/r5b_npt_fast_f32.log 2>&1
( TM_AMD_BAROSTAT_SLOW_PATH=1 timeout 300 python scripts/npt_bench.py f32 25 2000 2>&1 | tail -3 ) > gpurun_out/r5b_npt_slow_f32.log 2>&1
bash scripts/gpu_npt_trace.sh f64 > gpurun_out/r5b_npt_trace.log 2>&1
echo "== tests"; cat gpurun_out/r5b_tests.log
echo "== npt fast f64"; cat gpurun_out/r5b_npt_fast.log
echo "== npt two-launch f64"; cat gpurun_out/r5b_npt_two.log
echo "== npt slow f64"; cat gpurun_out/r5b_npt_slow.log
echo "== npt fast f32"; cat gpurun_out/r5b_npt_fast_f32.log
echo "== npt slow f32"; cat gpurun_out/r5b_npt_slow_f32.log
echo "== trace"; cat gpurun_out/r5b_npt_trace.log
