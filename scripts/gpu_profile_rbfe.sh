#!/bin/bash
# rocprofv3 kernel trace + stats of MD steps on the reference's RBFE state composition; prints the per-kernel table and the per-step
# table of the last 350 steps.  usage: scripts/gpu_profile_rbfe.sh <tag> <config4|config5> <f64|f32> [rbfe_steps.py args...]
# -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $ROOT/scripts/rbfe_steps.py "$@" > $OUT/run.log 2>&1
echo "rocprofv3 exit $?"
cd $ROOT
tail -1 $OUT/run.log | cut -c1-400
python - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
print("# rocprofv3 --kernel-trace --stats -- python scripts/rbfe_steps.py " + " ".join(sys.argv[2:]))
for f in glob.glob(out+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):9.0f} min={r['MinNs']:>8s} max={r['MaxNs']:>9s} pct={r['Percentage']}")
for f in glob.glob(out+'/**/*kernel_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    md=[r for r in rows if 'tmamd' in r['Kernel_Name']]
    upd=[i for i,r in enumerate(md) if 'k_update_forward_baoab' in r['Kernel_Name']]
    sel=md[upd[-351]+1:upd[-1]+1] if len(upd)>400 else md
    names=collections.Counter(); dur=collections.defaultdict(float); gap=0.0
    for a,b in zip(sel[:-1],sel[1:]):
        gap+=max(int(b['Start_Timestamp'])-int(a['End_Timestamp']),0)
    for r in sel:
        k=r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:60]
        names[k]+=1; dur[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    steps=sum(1 for r in sel if 'k_update_forward_baoab' in r['Kernel_Name'])
    wall=(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/steps/1e3
    print(f"-- the last {steps} MD steps of the trace: {wall:.2f} us per step; per step (us):")
    for k in sorted(dur,key=lambda k:-dur[k]):
        print(f"   {k:62s} {dur[k]/steps/1e3:7.2f}  ({names[k]/steps:.2f} launches/step, {dur[k]/names[k]/1e3:.2f} us each)")
    print(f"   idle between kernels                                           {gap/steps/1e3:7.2f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
