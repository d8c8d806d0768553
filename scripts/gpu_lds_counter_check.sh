#!/bin/bash
# Calibration of SQ_ACTIVE_INST_LDS: scripts/microbench/lds_atomics keeps every CU's LDS pipe saturated (16 waves per CU issuing nothing
# but LDS instructions); its counters (one PMC pass) against its kernel durations (one kernel-trace pass) say what fraction of a CU's
# cycles the counter reports for a saturated pipe -- the scale of `lds_busy` in the bench line.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/lds_check
mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -o pmc -- $ROOT/scripts/microbench/lds_atomics > $OUT/pmc.log 2>&1
echo "pmc exit $?"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- $ROOT/scripts/microbench/lds_atomics > $OUT/trace.log 2>&1
echo "trace exit $?"
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/lds_check'
pm = collections.defaultdict(dict)
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        pm[int(r['Dispatch_Id'])][r['Counter_Name']] = pm[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        pm[int(r['Dispatch_Id'])]['name'] = r['Kernel_Name'][:40]
tr = []
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    tr = [(r['Kernel_Name'][:40], 1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp']))) for r in rows]
ids = sorted(pm)
print("dispatch kernel dur_us insts_lds active_lds(quad) conflict  active*4/(256 CU * dur * 2.4 GHz)  active*4/(GRBM_GUI_ACTIVE/8 * 256)")
for k, i in enumerate(ids):
    if k >= len(tr): break
    c = pm[i]; dur = tr[k][1]
    a = c.get('SQ_ACTIVE_INST_LDS', 0.0)
    print(f"{i:4d} {tr[k][0][:24]:24s} {dur:9.1f} {c.get('SQ_INSTS_LDS',0):.4g} {a:.4g} {c.get('SQ_LDS_BANK_CONFLICT',0):.4g}   {4*a/(256*dur*2400):.3f}   {4*a/(c.get('GRBM_GUI_ACTIVE',1)/8*256):.3f}")
PY
