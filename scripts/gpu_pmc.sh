#!/bin/bash
# PMC counter passes (each in its own run, counters only -- never combined with sys/hip/hsa traces).
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
PREC=${PMC_PREC:-f64}
PMC=$ROOT/gpurun_out/pmc_md_$PREC
mkdir -p $PMC
CMD="python $ROOT/bench.py --steps ${PMC_STEPS:-200} --warmup 10 --no-cpu-baseline --no-npt --no-rc10 --no-rbfe-shape --profile-steps 0 --equil-scale 0.2 --equil-precision ${PMC_PREC:-f64} --precision ${PMC_PREC:-f64} ${BENCH_ARGS:-}"
cd /tmp
pass() { # name, counters...
  name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $PMC/$name -o $name -- $CMD > $PMC/$name.log 2>&1
  echo "pmc pass $name exit $?"
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $ROOT
python - $PMC $PREC <<'PY'
import csv, glob, collections, sys, json
traffic={}
for f in sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:44]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    disp=collections.Counter()
    seen=set()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:44]
        key=(k,r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); disp[k]+=1
    print('==',f)
    with open(f.replace('_counter_collection.csv','_summary.txt'),'w') as out:
        for k in sorted(agg, key=lambda k:-disp[k]):
            line=f"{k:46s} n={disp[k]:5d} "+' '.join(f"{c}={v/disp[k]:.5g}" for c,v in sorted(agg[k].items()))
            out.write(line+"\n")
            if 'tiles' in k or 'find_ixns' in k or 'baoab' in k: print(line)
            # the MD-step variant of the tile kernel: forces only (<Real, false, true, false>)
            if k.startswith('k_nonbonded_tiles<%s, false, true, false' % {'f64': 'double', 'f32': 'float'}[sys.argv[2]]) and disp[k] > 50:
                for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT',
                          'SQ_ACTIVE_INST_LDS', 'SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'):
                    if c in agg[k]: traffic[c] = agg[k][c] / disp[k]
if 'FETCH_SIZE' in traffic and 'WRITE_SIZE' in traffic:
    # KB per dispatch; the guide's gfx950 correction: FETCH_SIZE tallies 128-B read requests at 64 B -> double it
    b = (2.0 * traffic['FETCH_SIZE'] + traffic['WRITE_SIZE']) * 1024
    rec = {"bytes": b, "fetch_kb_raw": traffic['FETCH_SIZE'], "write_kb_raw": traffic['WRITE_SIZE'],
           "correction": "2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)", "source": "profiles/pmc_md_%s.txt" % sys.argv[2],
           # per-dispatch averages of the SQ passes of the same command (instruction counts; cycle counters in quad-cycles)
           "sq": {k: v for k, v in traffic.items() if k.startswith('SQ_') or k.startswith('GRBM')},
           # which build of the library these counters belong to (csrc/.build_stamp: hash of sources + flags); bench.py prints
           # traffic_stale: true when it times another build
           "build_stamp": open('timemachine_amd/csrc/.build_stamp').read().strip()}
    json.dump({sys.argv[2]: rec}, open(sys.argv[1] + '/pmc_traffic.json', 'w'))
    print('traffic bytes per launch', b)
PY
find $PMC -name "*counter_collection.csv" -delete; find $PMC -name "*.db" -delete; du -sh gpurun_out
