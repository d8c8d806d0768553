#!/bin/bash
# round 4, first look at the row-block kernel: targeted parity tests with the kernel forced on every forces-only launch, then
# an A/B bench (item kernel vs row-block kernel), then the whole GPU suite with the kernel forced
set -u
mkdir -p gpurun_out
L=gpurun_out/r4a.log
: > $L
export TM_AMD_ROWBLOCK_MIN_K=0
echo "== targeted tests, row-block kernel forced" >> $L
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nonbonded_golden or config2_all_terms or config1_both or dhfr" >> $L 2>&1
echo "rc=$?" >> $L
timeout 400 python -m pytest tests/test_gpu_nonbonded_cases.py -m gpu -x -q >> $L 2>&1
echo "rc=$?" >> $L
unset TM_AMD_ROWBLOCK_MIN_K
for mk in 1000000000 8192; do
  export TM_AMD_ROWBLOCK_MIN_K=$mk
  echo "== bench, TM_AMD_ROWBLOCK_MIN_K=$mk" >> $L
  timeout 240 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-npt 2>>$L | tail -1 > gpurun_out/r4a_bench_$mk.json
  python - >> $L 2>&1 <<PY
import json
d=json.load(open("gpurun_out/r4a_bench_$mk.json"))
print({k:d.get(k) for k in ('value','ms_per_step','ns_day_f32')}, 'tile_ms', d['roofline'].get('kernel_ms'), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('rc1.0')})
PY
done
unset TM_AMD_ROWBLOCK_MIN_K
echo "== whole GPU suite, row-block kernel forced" >> $L
TM_AMD_ROWBLOCK_MIN_K=0 timeout 900 python -m pytest tests -m gpu -x -q >> $L 2>&1
echo "rc=$?" >> $L
tail -60 $L
