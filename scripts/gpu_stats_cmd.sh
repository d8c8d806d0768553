#!/bin/bash
# rocprofv3 kernel stats of an arbitrary command: scripts/gpu_stats_cmd.sh <tag> <top-N> <command...>  -> gpurun_out/prof_<tag>/ and a table on stdout
set -u
TAG=$1; TOP=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/run.log 2>&1
echo "# rocprofv3 --kernel-trace --stats -- $* (exit $?)"
cd $ROOT
python - "$OUT" "$TOP" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2])]:
        print(r["Name"][:110].ljust(110), r["Calls"].rjust(7), "avg %9.1f us  min %8.1f  max %9.1f  %s%%" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
