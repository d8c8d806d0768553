#!/bin/bash
# round 5 artifacts: everything profiles/r05_* holds, on the final library (one gpurun call)
set -u
TAG=${1:-r05v1}
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
bash scripts/gpu_round_artifacts.sh $TAG > gpurun_out/art_$TAG.log 2>&1
A=gpurun_out/art_$TAG
( timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) > $A/gpu_tests.txt 2>&1
( TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_guard.so timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 ) > $A/gpu_tests_guard.txt 2>&1
bash scripts/gpu_npt_trace.sh f64 > $A/npt_trace_f64.txt 2>&1
( timeout 300 python scripts/tile_ablate.py 2>&1 | tail -2 ) > $A/tile_frame.txt 2>&1
( TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/libtimemachine_amd_tbatch.so timeout 300 python scripts/batch_timeline.py 2>&1 | tail -12 ) > $A/batch_timeline.txt 2>&1
( scripts/microbench/persistent_step; scripts/microbench/fork_join; scripts/microbench/lds_atomics | head -8 ) > $A/microbench.txt 2>&1
( timeout 300 python scripts/host_cpu_probe.py 2>&1 | tail -3; TM_AMD_SPIN_WAIT=1 timeout 300 python scripts/host_cpu_probe.py 2>&1 | tail -3 ) > $A/host_cpu.txt 2>&1
( timeout 300 python scripts/host_cpu_probe3.py 2>&1 | tail -2; echo '-- TM_AMD_SPIN_WAIT=1 (no run-ahead bound, spinning waits)'; TM_AMD_SPIN_WAIT=1 timeout 300 python scripts/host_cpu_probe3.py 2>&1 | tail -2 ) > $A/host_cpu_call_length.txt 2>&1
( env -u GPU_MAX_HW_QUEUES TM_AMD_BINDING=ctypes GROUP_COUNTS=1,2,3,4 timeout 300 python scripts/group_bench.py f64 1500 2>&1 | tail -5 ) > $A/group_ctypes.txt 2>&1
( timeout 300 python scripts/size_sweep.py f32 2>&1 | tail -7; timeout 300 python scripts/size_sweep.py f64 2>&1 | tail -7 ) > $A/size_sweep.txt 2>&1
tail -30 gpurun_out/art_$TAG.log
for f in gpu_tests gpu_tests_guard npt_trace_f64 tile_frame batch_timeline host_cpu host_cpu_call_length group_ctypes size_sweep; do echo "== $f"; tail -12 $A/$f.txt; done
