#!/bin/bash
# rocprofv3 kernel trace of four replicas stepped together (custom_ops.multiple_steps_group): how much of the launches' time overlaps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_group
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
GROUP_COUNTS=${GROUP_COUNTS:-4} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o g -- python $ROOT/scripts/group_bench.py ${1:-f64} 800 > $OUT/run.log 2>&1
cd $ROOT
tail -2 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'tmamd' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    tiles = [i for i, r in enumerate(rows) if 'k_nonbonded_tiles' in r['Kernel_Name'] and 'false, true, false' in r['Kernel_Name']]
    sel = rows[tiles[-1601]:tiles[-1]]  # the last 1600 force launches = 400 rounds of four replicas
    t0, t1 = int(sel[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in sel)
    span = (t1 - t0) / 1e3
    dur = collections.defaultdict(float); cnt = collections.Counter()
    for r in sel:
        k = r['Kernel_Name'].split('(')[0].replace('void tmamd::', '')[:44]
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[k] += 1
    n_tiles = sum(1 for r in sel if 'k_nonbonded_tiles' in r['Kernel_Name'])
    # time with >= 1 / >= 2 kernels in flight (sweep over start / end events)
    ev = sorted([(int(r['Start_Timestamp']), 1) for r in sel] + [(int(r['End_Timestamp']), -1) for r in sel])
    depth = 0; last = ev[0][0]; busy = [0.0, 0.0, 0.0, 0.0, 0.0]
    for t, d in ev:
        busy[min(depth, 4)] += t - last; last = t; depth += d
    tot = sum(busy)
    lines = [f"-- four replicas stepped together, {n_tiles} force launches: {span / n_tiles:.2f} us of wall time per replica-step; kernel time summed over the launches {sum(dur.values()) / n_tiles:.2f} us per replica-step",
             "   time with 0 / 1 / 2 / 3 / >= 4 kernels in flight: " + " / ".join(f"{100 * b / tot:.1f} %" for b in busy)]
    for k in sorted(dur, key=lambda k: -dur[k]):
        lines.append(f"   {k:46s} {dur[k] / cnt[k]:8.2f} us per launch, {cnt[k] / n_tiles:.2f} launches per replica-step")
    open(out + '/group_trace.txt', 'w').write("\n".join(lines) + "\n")
    print("\n".join(lines))
PY
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
