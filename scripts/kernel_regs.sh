#!/bin/bash
# register / spill / LDS report for the kernels of one source file: kernel_regs.sh nonbonded.hip [name filter] [-D...]
src=$1; filt=${2:-.}; shift; shift
cd "$(dirname "$0")/../timemachine_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -I../../include \
  -Rpass-analysis=kernel-resource-usage -x hip -c $src -o /tmp/_regs_$$.o 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0].replace('void tmamd::','')}; rows.append(cur); continue
    m=re.search(r'remark: (?:[^:]+:\d+:\d+: )?\s*([A-Za-z ]+[A-Za-z])\s*(?:\[bytes/lane\]|\[bytes/block\]|\[waves/SIMD\])?: (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows:
    if re.search(r'$filt', r['name']):
        print(f\"{r['name'][:70]:70s} vgpr={r.get('VGPRs','?'):>4s} agpr={r.get('AGPRs','?'):>3s} sgpr={r.get('TotalSGPRs','?'):>4s} spillV={r.get('VGPR Spill','?'):>3s} spillS={r.get('SGPR Spill','?'):>3s} scratch={r.get('ScratchSize','?'):>4s} occ={r.get('Occupancy','?'):>2s} lds={r.get('LDS Size','?')}\")
"
