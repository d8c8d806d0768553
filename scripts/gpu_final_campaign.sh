#!/bin/bash
# the closing campaign of round 6 on the final library: many more seeds of every randomized layer (one gpurun call).
set -u
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
A=gpurun_out/final_campaign; mkdir -p $A
( echo "# python scripts/fuzz_campaign_long.py 800 20000   (interleavings: every fast path on = off, 16 runs per seed)"
  timeout 2400 python scripts/fuzz_campaign_long.py 800 20000 2>&1 | grep -v amdgpu.ids
  echo "# python scripts/fuzz_campaign_config5.py 1200 20000 1200   (the same, single windows at config-5 size)"
  timeout 1500 python scripts/fuzz_campaign_config5.py 1200 20000 1200 2>&1 | grep -v amdgpu.ids
  echo "# python scripts/fuzz_parity.py 20000 1500   (random systems against the oracle)"
  timeout 1200 python scripts/fuzz_parity.py 20000 1500 2>&1 | grep -v amdgpu.ids ) > $A/campaign.txt 2>&1
grep -v "^\.\.\." $A/campaign.txt | tail -30
