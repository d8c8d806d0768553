#!/bin/bash
# list-build probe: for each library variant given (paths relative to the repo root; "product" = the built library), a short
# MD run of the bench workload under rocprofv3 --kernel-trace; prints per kernel calls / average, and k_find_ixns split into
# rebuilding launches and launches that only read the flag.   usage: scripts/gpu_nbl_probe.sh <f64|f32> product [variant.so ...]
set -u
PREC=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
for V in "$@"; do
  TAG=$(basename $V .so)
  OUT=$ROOT/gpurun_out/nblprobe_${TAG}_$PREC
  rm -rf $OUT; mkdir -p $OUT
  if [ "$V" != product ]; then export TM_AMD_LIB=$ROOT/$V; else unset TM_AMD_LIB; fi
  EXTRA=""; [ "$PREC" = f32 ] && EXTRA="--precision f32"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-npt --no-rc10 --no-rbfe-shape --profile-steps 0 $EXTRA > $OUT/run.log 2>&1)
  python - "$OUT" "$TAG" "$PREC" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
real = 'k_nonbonded_tiles<float' if sys.argv[3] == 'f32' else 'k_nonbonded_tiles<double'
for line in open(out + '/run.log'):
    if line.startswith('{'):
        d = json.loads(line); print(tag, 'ns/day', d['value'], 'ms/step', d['ms_per_step'])
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    md = [r for r in rows if 'tmamd' in r['Kernel_Name']]
    tiles = [i for i, r in enumerate(md) if real in r['Kernel_Name'] and 'false, true, false' in r['Kernel_Name']]
    sel = md[tiles[-351]:tiles[-1]] if len(tiles) > 400 else md
    dur = collections.defaultdict(list)
    for r in sel:
        k = r['Kernel_Name'].split('(')[0].replace('void tmamd::', '')[:44]
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    steps = sum(len(v) for k, v in dur.items() if 'k_nonbonded_tiles' in k)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:44s} n={len(v):5d} avg={sum(v)/len(v):7.2f} us  per step {sum(v)/steps:6.2f}")
        if 'k_find_ixns' in k:
            big = [x for x in v if x > 15.0]; small = [x for x in v if x <= 15.0]
            if big and small:
                print(f"      rebuilds n={len(big)} avg={sum(big)/len(big):.2f} min={min(big):.2f} max={max(big):.2f};  flag-only n={len(small)} avg={sum(small)/len(small):.2f}")
PY
  find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
done
