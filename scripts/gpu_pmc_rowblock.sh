#!/bin/bash
# SQ counters of the row-block kernel (and of the item kernel on the same frames), per launch: rocprofv3 --pmc passes (counters only)
# of scripts/rb_frame_bench.py.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_rowblock
mkdir -p $OUT
cd /tmp
pass() { name=$1; shift; RB_MIN_KS=0,2000000000 timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/scripts/rb_frame_bench.py > $OUT/$name.log 2>&1; echo "pass $name exit $?"; }
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
pass sq2 SQ_INSTS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM
pass mem FETCH_SIZE WRITE_SIZE
cd $ROOT
python - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in sorted(glob.glob(sys.argv[1] + '/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void tmamd::', '')[:52]
        if 'k_nonbonded_rowblocks' in k or 'k_nonbonded_tiles' in k:
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
lines = []
for k in sorted(agg):
    lines.append(k)
    for c, v in sorted(agg[k].items()):
        lines.append(f"    {c:24s} {v / max(len(disp[(k, c)]), 1):14.5g}   per launch ({len(disp[(k, c)])} launches)")
open(sys.argv[1] + '/summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
