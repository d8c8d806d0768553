#!/bin/bash
# rocprofv3 PMC passes over scripts/ablate.py's worker (20 forces-only launches of the tile kernel on a fixed, equilibrated
# DHFR-sized frame) for one library variant.  usage: scripts/pmc_tiles.sh <tag> [lib.so]   -> gpurun_out/pmc_<tag>/summary.txt
# Counters only (never combined with sys/hip/hsa traces).
set -u
TAG=$1; LIB=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$LIB" ] && export TM_AMD_LIB=$LIB
[ -f /tmp/ablate_frame.npy ] || python $ROOT/scripts/ablate.py frame > /dev/null 2>&1
cd /tmp
pass() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/scripts/ablate.py worker > $OUT/$name.log 2>&1
  echo "pmc pass $name exit $?"; }
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE
pass c SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_MISC
pass d SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES
[ -n "${PMC_SKIP_TRAFFIC:-}" ] || pass fetch FETCH_SIZE
[ -n "${PMC_SKIP_TRAFFIC:-}" ] || pass write WRITE_SIZE
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out=sys.argv[1]
lines=[]
for f in sorted(glob.glob(out+'/*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set(); disp=collections.Counter()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void tmamd::','')[:48]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        key=(k,r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); disp[k]+=1
    for k in sorted(agg, key=lambda k:-disp[k]):
        if 'tiles' in k:
            lines.append(f"{k:50s} n={disp[k]:4d} "+' '.join(f"{c}={v/disp[k]:.5g}" for c,v in sorted(agg[k].items())))
open(out+'/summary.txt','w').write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
