#!/bin/bash
# kernel-by-kernel view of barostat attempts: rocprofv3 kernel trace of scripts/npt_bench.py, then the launches between two
# consecutive k_barostat_* decisions.  gpu_npt_trace.sh <f32|f64>
set -u
prec=${1:-f32}
tag=npt_$prec
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/scripts/npt_bench.py $prec 25 500 > $GRAFT_REPO_ROOT/gpurun_out/$tag/run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/$tag/run.log
python - "$tag" <<'PY'
import csv, sys, collections
tag=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:60]
# the last attempts of the run: find launches of the barostat's decision kernel
idx=[i for i,r in enumerate(rows) if 'barostat' in r['Kernel_Name'].lower() and 'decide' in r['Kernel_Name'].lower()]
print("decide launches", len(idx))
if len(idx)>=3:
    a,b=idx[-3],idx[-2]
    # one period: from just after decision a to decision b (25 MD steps + one attempt)
    seg=rows[a+1:b+1]
    t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
    print(f"period wall {1e-3*(t1-t0):.1f} us, {len(seg)} launches")
    agg=collections.OrderedDict()
    for r in seg:
        k=nm(r); d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        c=agg.setdefault(k,[0,0]); c[0]+=1; c[1]+=d
    for k,(n,d) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
        print(f"  {k:62s} n={n:3d} total={1e-3*d:8.1f} us avg={1e-3*d/n:7.1f}")
    # the attempt itself: launches from the first barostat kernel before decision b to b, in order
    j=b
    while j>a and not ('barostat' in rows[j]['Kernel_Name'].lower() and 'decide' not in rows[j]['Kernel_Name'].lower() and 'propose' in rows[j]['Kernel_Name'].lower()): j-=1
    # every attempt of the run: its launches from the proposal kernel to the decision, and how often the list was rebuilt inside it
    props=[i for i,r in enumerate(rows) if 'barostat' in r['Kernel_Name'].lower() and 'propose' in r['Kernel_Name'].lower()]
    durs=[]; rebuilds=0
    for p0 in props[5:]:
        q=p0
        while q < len(rows) and 'decide' not in rows[q]['Kernel_Name'].lower(): q+=1
        if q >= len(rows): break
        durs.append(1e-3*(int(rows[q]['End_Timestamp'])-int(rows[p0]['Start_Timestamp'])))
        rebuilds += any('find_ixns' in rows[k]['Kernel_Name'] and int(rows[k]['End_Timestamp'])-int(rows[k]['Start_Timestamp']) > 20000 for k in range(p0,q))
    if durs:
        import statistics
        print(f"attempts {len(durs)}: proposal .. decision {statistics.mean(durs):.1f} us mean, {statistics.median(durs):.1f} median; list rebuilt inside {rebuilds} of them")
    print("attempt launches in order:")
    tA=int(rows[j]['Start_Timestamp'])
    for r in rows[j:b+4]:
        print(f"  +{1e-3*(int(r['Start_Timestamp'])-tA):8.1f} us  {1e-3*(int(r['End_Timestamp'])-int(r['Start_Timestamp'])):7.1f} us  {nm(r)}")
PY
rm -f gpurun_out/$tag/*.db gpurun_out/$tag/*kernel_trace.csv
