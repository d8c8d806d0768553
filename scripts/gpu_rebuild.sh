#!/bin/bash
# neighbor-list build kernel durations for library variants: gpu_rebuild.sh <lib|default> ...
set -u
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset TM_AMD_LIB; else export TM_AMD_LIB=$GRAFT_REPO_ROOT/timemachine_amd/csrc/$lib; fi
  tag=rb_$(echo $lib | tr -c 'a-zA-Z0-9' '_')
  mkdir -p gpurun_out/$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-npt --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1)
  echo "== $lib: $(tail -1 gpurun_out/$tag/bench.log | cut -c1-120)"
  python - "$tag" <<'PY'
import csv, sys
tag=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_stats.csv')))
for r in rows:
    n=r['Name'].split('(')[0].replace('void tmamd::','')[:48]
    if 'find_ixns' in n or 'block_bounds' in n or 'check_gather' in n or 'baoab' in n:
        print(f"   {n:44s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:7.1f}us max={float(r['MaxNs'])/1e3:7.1f}")
PY
  rm -rf gpurun_out/$tag
done
