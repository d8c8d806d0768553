#!/bin/bash
# Do two independent MD processes on ONE GPU add up to more than one?  (kernel boundaries and the tile kernel's tail of one replica
# filled by the other's work)  usage: gpu_two_procs.sh [n_procs=2] [extra bench args]
set -u
N=${1:-2}; shift || true
mkdir -p gpurun_out
L=gpurun_out/two_procs.log
: > $L
echo "== one process" >> $L
timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-npt --no-rc10 --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step')})" >> $L
echo "== $N processes at once" >> $L
for i in $(seq 1 $N); do
  (timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-npt --no-rc10 --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('value','ms_per_step')})" >> $L) &
done
wait
cat $L
