#!/bin/bash
# One gpurun call: smoke -> parity tests -> short bench -> rocprofv3 kernel stats.  Everything is logged under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>&1 | head -8
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -15 gpurun_out/smoke.log
echo "== pytest"; timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --timeout 300 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|Error|error" gpurun_out/pytest_gpu.log | head -60; tail -5 gpurun_out/pytest_gpu.log
if [ "${RUN_BENCH:-1}" = "1" ]; then
  echo "== bench"; timeout 600 python bench.py --steps ${BENCH_STEPS:-1000} --warmup 200 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -5 gpurun_out/bench.log
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  echo "== rocprofv3"; cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-npt --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "rocprof exit $?"; cd $GRAFT_REPO_ROOT
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
fi
find gpurun_out/prof -name "*.db" -delete 2>/dev/null; du -sh gpurun_out
