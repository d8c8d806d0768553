#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; prints the per-kernel table and the idle time between kernels.
# usage: scripts/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-npt --no-rc10 --no-rbfe-shape --profile-steps 0 "$@" > $OUT/run.log 2>&1
cd $ROOT
python - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for f in glob.glob(out+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    with open(out+'/kernel_stats.txt','w') as fh:
        for r in rows[:14]:
            line=f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):9.0f} min={r['MinNs']:>8s} max={r['MaxNs']:>9s} pct={r['Percentage']}"
            print(line); fh.write(line+"\n")
for f in glob.glob(out+'/**/*kernel_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # the last 400 x 4 launches = the timed MD steps of the f64 run
    md=[r for r in rows if 'tmamd' in r['Kernel_Name']]
    names=collections.Counter(); dur=collections.defaultdict(float); gap=0.0; n=0
    # the timed MD steps: the last 350 forces-only tile launches of the benchmarked precision (the other precision's short
    # leg may follow them in the trace)
    real='k_nonbonded_tiles<float, false, true, false' if 'f32' in sys.argv[2:] else 'k_nonbonded_tiles<double, false, true, false'
    tiles=[i for i,r in enumerate(md) if real in r['Kernel_Name']]
    names=collections.Counter(); dur=collections.defaultdict(float); gap=0.0; n=0
    sel=md[tiles[-351]:tiles[-1]] if len(tiles)>400 else md
    for a,b in zip(sel[:-1],sel[1:]):
        g=int(b['Start_Timestamp'])-int(a['End_Timestamp'])
        gap+=max(g,0); n+=1
    for r in sel:
        k=r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:40]
        names[k]+=1; dur[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    steps=sum(1 for r in sel if 'k_nonbonded_tiles' in r['Kernel_Name'])
    print(f"-- the last {steps} timed MD steps of the trace; per step (us):")
    for k in sorted(dur,key=lambda k:-dur[k]):
        print(f"   {k:42s} {dur[k]/steps/1e3:7.2f}  ({names[k]/steps:.2f} launches/step, {dur[k]/names[k]/1e3:.2f} us each)")
    print(f"   idle between kernels                       {gap/steps/1e3:7.2f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
