#!/bin/bash
# rocprofv3 kernel stats for one bench configuration: gpu_stats.sh <tag> [bench args...]
set -u
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-npt --no-rbfe-shape --profile-steps 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.log 2>&1
echo "rocprof $tag exit $?"
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/$tag/bench.log | cut -c1-200
python - "$tag" <<'PY'
import csv, sys
tag=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_stats.csv')))
for r in rows[:22]:
    n=r['Name'].split('(')[0].replace('void tmamd::','')[:48]
    print(f"{n:50s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f}us min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
rm -f gpurun_out/$tag/*.db gpurun_out/$tag/*kernel_trace.csv
