#!/bin/bash
# per-kernel durations and the gaps between launches for small systems: gpu_size_trace.sh <f32|f64>
set -u
prec=${1:-f32}
tag=size_$prec
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/scripts/size_sweep.py $prec > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
python - "$tag" <<'PY'
import csv, sys, collections
tag=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:58]
# split the trace into the six systems: by grid size of the update kernel (ceil(N/64) workgroups of 64)
segs=collections.OrderedDict()
cur=None
for i,r in enumerate(rows):
    if 'k_update_forward_baoab' in r['Kernel_Name']:
        cur=int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size',0))
    if cur is not None:
        segs.setdefault(cur,[]).append(r)
for g,seg in segs.items():
    seg=seg[-3000:]  # the tail: the timed production run of that system
    t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
    n_upd=sum(1 for r in seg if 'k_update_forward_baoab' in r['Kernel_Name'])
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
    print(f"grid {g}: {n_upd} steps, wall/step {1e-3*(t1-t0)/max(n_upd,1):.2f} us, busy/step {1e-3*busy/max(n_upd,1):.2f} us")
    agg=collections.OrderedDict()
    for r in seg:
        c=agg.setdefault(nm(r),[0,0]); c[0]+=1; c[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    for k,(n,d) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:7]:
        print(f"    {k:60s} n={n:5d} avg={1e-3*d/n:7.2f} us  per step {1e-3*d/max(n_upd,1):6.2f}")
PY
rm -f gpurun_out/$tag/*.db gpurun_out/$tag/*kernel_trace.csv
